#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/ from the read-only reference checkout.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite use the committed
outputs.  Inputs:
  /root/reference/examples/nano_hits.parquet                  (24,586 rows, 2 row groups; ClickBench `hits` sample)
  /root/reference/benchmark/tpch/data/sf0.001/lineitem.parquet (6,005 rows)
Outputs:
  nano_hits_cols.parquet   the string / integer columns the hot path's configs use (zstd)
  lineitem_sf0001.parquet  l_shipdate, l_discount, l_quantity, l_extendedprice
  expected.json            answers computed by pyarrow (Arrow semantics == the reference's generic path,
                           liquid_array/mod.rs:265-280) + the SQL-level answers pinned by the reference's own
                           snapshots (src/datafusion-local/src/tests/snapshots: URL LIKE 'https://%' -> 23113 rows,
                           WatchID = 6978470580070504163 -> 1 row, URL LIKE '%tours%' -> 11 rows)
"""
import hashlib
import json
import os

import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.parquet as pq

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def mask_digest(mask: pa.ChunkedArray) -> str:
    b = pc.fill_null(mask, False).combine_chunks().to_numpy(zero_copy_only=False)
    return hashlib.sha256(bytes(b.astype("u1"))).hexdigest()[:16]


def main():
    hits = pq.read_table(os.path.join(REF, "examples/nano_hits.parquet"))
    cols = ["URL", "SearchPhrase", "WatchID", "UserID", "RegionID", "EventDate", "ResolutionWidth", "AdvEngineID",
            "EventTime", "CounterID"]
    hits = hits.select(cols)
    pq.write_table(hits, os.path.join(HERE, "nano_hits_cols.parquet"), compression="zstd", compression_level=19,
                   row_group_size=24576)
    li = pq.read_table(os.path.join(REF, "benchmark/tpch/data/sf0.001/lineitem.parquet"))
    li = li.select(["l_shipdate", "l_discount", "l_quantity", "l_extendedprice"])
    pq.write_table(li, os.path.join(HERE, "lineitem_sf0001.parquet"), compression="zstd", compression_level=19)

    exp = {"nano_hits_rows": hits.num_rows, "lineitem_rows": li.num_rows, "predicates": []}

    def add(table_name, table, col, op, literal, mask):
        exp["predicates"].append({"table": table_name, "column": col, "op": op, "literal": literal,
                                  "count": int(pc.sum(pc.fill_null(mask, False)).as_py() or 0),
                                  "digest": mask_digest(mask)})

    cmp = {"eq": pc.equal, "ne": pc.not_equal, "lt": pc.less, "le": pc.less_equal, "gt": pc.greater,
           "ge": pc.greater_equal}
    for col, lits in (("WatchID", [12, 6978470580070504163, 4611686071420045196]),
                      ("UserID", [12, 0, 435090932899640449, -1]),
                      ("RegionID", [229, 2, 100000]), ("ResolutionWidth", [1368, 0, 1920]),
                      ("AdvEngineID", [0, 2]), ("EventDate", [15900, 15901]), ("CounterID", [62])):
        for op, fn in cmp.items():
            for lit in lits:
                add("nano_hits", hits, col, op, lit, fn(hits[col], pa.scalar(lit, hits[col].type)))
    for needle in ("google", "tours", "yandex", "https://", "&", "page=", "zzzz"):
        add("nano_hits", hits, "URL", "like", "%" + needle + "%", pc.match_substring(hits["URL"], needle))
        add("nano_hits", hits, "URL", "not_like", "%" + needle + "%", pc.invert(pc.match_substring(hits["URL"], needle)))
    for op, fn in cmp.items():
        for lit in ("", "http://kinopoisk.ru", "http://zzz", hits["URL"][100].as_py(), hits["URL"][100].as_py()[:20]):
            add("nano_hits", hits, "URL", op, lit, fn(hits["URL"], pa.scalar(lit)))
        for lit in ("", hits["SearchPhrase"][5].as_py()):
            add("nano_hits", hits, "SearchPhrase", op, lit, fn(hits["SearchPhrase"], pa.scalar(lit)))
    add("nano_hits", hits, "URL", "like_prefix", "https://%", pc.match_like(hits["URL"], "https://%"))
    import datetime
    import decimal
    for op, fn in cmp.items():
        for d in (datetime.date(1994, 1, 1), datetime.date(1995, 1, 1), datetime.date(1998, 12, 1)):
            add("lineitem", li, "l_shipdate", op, d.isoformat(), fn(li["l_shipdate"], pa.scalar(d, li["l_shipdate"].type)))
        for v in ("0.05", "0.07", "0.00"):
            add("lineitem", li, "l_discount", op, v,
                fn(li["l_discount"], pa.scalar(decimal.Decimal(v), li["l_discount"].type)))
    exp["sql_goldens"] = {  # pinned by the reference's datafusion-local snapshots
        "URL LIKE 'https://%'": 23113, "WatchID = 6978470580070504163": 1, "URL LIKE '%tours%'": 11}
    exp["tours_urls"] = [u for u in hits["URL"].to_pylist() if "tours" in u]
    with open(os.path.join(HERE, "expected.json"), "w") as f:
        json.dump(exp, f, indent=0, sort_keys=True)
    print("wrote", len(exp["predicates"]), "predicate answers")


if __name__ == "__main__":
    main()
