"""The pushed-down predicates of the 43 ClickBench queries (reference: benchmark/clickbench/queries/q0.sql .. q42.sql) and
a synthetic `hits`-shaped table to run them on (BASELINE.json config 5; SURVEY.md §8d "Config 5").

Only the WHERE clause of a query reaches the cache (`LiquidCacheReader::build_predicate_filter`); 24 of the 43 queries
have one.  `QUERIES[n]` lists the conjuncts of q<n>.sql as the reference's conjunct split produces them
(row_filter.rs:428-515); the evaluation order is `pushdown.LiquidRowFilter`'s.  Two translations are made on the host,
where the reference unwraps casts around the column (liquid_expr.rs:150-164):
  * `"EventDate"::INT::DATE >= '2013-07-01'` — EventDate is a UInt16 day number in `hits`: compared with the literal's day
    number (15887 = 2013-07-01);
  * `"TraficSourceID" IN (-1, 6)` — an IN list is not a LiquidExpr in the reference (it materialises the column and lets
    Arrow evaluate); here it is the Kleene OR of two equality passes over the same column, which gives the same mask.
There is no dataset access on the benchmark machines (hits.parquet is a 14 GB download), hence synthetic columns with the
types of the reference's examples/nano_hits.parquet schema and value distributions that keep every predicate selective
the way it is on the real table (AdvEngineID <> 0: ~5 %; SearchPhrase <> '': ~13 %; CounterID = 62: ~0.7 %, ...).
"""
from __future__ import annotations

import datetime
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Union

import numpy as np
import pyarrow as pa

from . import _native as N
from .cache import CacheExpression, LiquidCache, ParquetArrayID
from .pushdown import AnyOf, Column, Conjunct

_EPOCH = datetime.date(1970, 1, 1)


def _days(s: str) -> int:
    return (datetime.date.fromisoformat(s) - _EPOCH).days


def _july(lo: str, hi: str) -> List[Conjunct]:
    return [Conjunct("EventDate", ">=", _days(lo)), Conjunct("EventDate", "<=", _days(hi))]


_Q36_COMMON = [Conjunct("CounterID", "=", 62)] + _july("2013-07-01", "2013-07-31")

#: query index -> conjuncts of its WHERE clause (queries without a WHERE clause push nothing down)
QUERIES: Dict[int, List[Union[Conjunct, AnyOf]]] = {
    1: [Conjunct("AdvEngineID", "!=", 0)],
    7: [Conjunct("AdvEngineID", "!=", 0)],
    10: [Conjunct("MobilePhoneModel", "!=", "")],
    11: [Conjunct("MobilePhoneModel", "!=", "")],
    12: [Conjunct("SearchPhrase", "!=", "")],
    13: [Conjunct("SearchPhrase", "!=", "")],
    14: [Conjunct("SearchPhrase", "!=", "")],
    19: [Conjunct("UserID", "=", 435090932899640449)],
    20: [Conjunct("URL", "like", "%google%")],
    21: [Conjunct("URL", "like", "%google%"), Conjunct("SearchPhrase", "!=", "")],
    22: [Conjunct("Title", "like", "%Google%"), Conjunct("URL", "not like", "%.google.%"), Conjunct("SearchPhrase", "!=", "")],
    23: [Conjunct("URL", "like", "%google%")],
    24: [Conjunct("SearchPhrase", "!=", "")],
    25: [Conjunct("SearchPhrase", "!=", "")],
    26: [Conjunct("SearchPhrase", "!=", "")],
    27: [Conjunct("URL", "!=", "")],
    28: [Conjunct("Referer", "!=", "")],
    30: [Conjunct("SearchPhrase", "!=", "")],
    31: [Conjunct("SearchPhrase", "!=", "")],
    36: _Q36_COMMON + [Conjunct("DontCountHits", "=", 0), Conjunct("IsRefresh", "=", 0), Conjunct("URL", "!=", "")],
    37: _Q36_COMMON + [Conjunct("DontCountHits", "=", 0), Conjunct("IsRefresh", "=", 0), Conjunct("Title", "!=", "")],
    38: _Q36_COMMON + [Conjunct("IsRefresh", "=", 0), Conjunct("IsLink", "!=", 0), Conjunct("IsDownload", "=", 0)],
    39: _Q36_COMMON + [Conjunct("IsRefresh", "=", 0)],
    40: _Q36_COMMON + [Conjunct("IsRefresh", "=", 0),
                       AnyOf([Conjunct("TraficSourceID", "=", -1), Conjunct("TraficSourceID", "=", 6)]),
                       Conjunct("RefererHash", "=", 3594120000172545465)],
    41: _Q36_COMMON + [Conjunct("IsRefresh", "=", 0), Conjunct("DontCountHits", "=", 0),
                       Conjunct("URLHash", "=", 2868770270353813622)],
    42: [Conjunct("CounterID", "=", 62)] + _july("2013-07-14", "2013-07-15") +
        [Conjunct("IsRefresh", "=", 0), Conjunct("DontCountHits", "=", 0)],
}
N_QUERIES = 43

#: column name -> (ParquetArrayID column id, Arrow type, expression hint) — ids follow the `hits` schema order
SCHEMA = {
    "Title": (2, pa.string(), CacheExpression.SUBSTRING_SEARCH), "EventDate": (5, pa.uint16(), None),
    "CounterID": (6, pa.int32(), None), "UserID": (9, pa.int64(), None), "URL": (13, pa.string(), CacheExpression.SUBSTRING_SEARCH),
    "Referer": (14, pa.string(), None), "IsRefresh": (15, pa.int16(), None), "MobilePhoneModel": (34, pa.string(), None),
    "TraficSourceID": (37, pa.int16(), None), "SearchPhrase": (39, pa.string(), None), "AdvEngineID": (40, pa.int16(), None),
    "IsLink": (52, pa.int16(), None), "IsDownload": (53, pa.int16(), None), "DontCountHits": (61, pa.int16(), None),
    "RefererHash": (102, pa.int64(), None), "URLHash": (103, pa.int64(), None),
}
_STRINGS = ("Title", "URL", "Referer", "MobilePhoneModel", "SearchPhrase")


def synth_batch(name: str, seed: int, b: int, rows: int, scratch=None) -> pa.Array:
    """One 8192-row batch of synthetic column `name`; deterministic in (seed, b)."""
    L = N.load()
    cid, dtype, _ = SCHEMA[name]
    if name in _STRINGS:
        offs = np.zeros(rows + 1, np.int32)
        data = np.zeros(rows * 512, np.uint8) if scratch is None else scratch
        s = seed * 131 + cid
        if name == "URL":
            n = N.load_bench().lc_synth_url_batch(s, b, rows, min(2200, rows), 159, offs.ctypes.data, data.ctypes.data, data.size)
        elif name == "Referer":  # URL shaped; about one row in five has no referer
            n = N.load_bench().lc_synth_url_batch(s, b, rows, min(1500, rows), 400, offs.ctypes.data, data.ctypes.data, data.size)
            blank = np.random.default_rng([seed, cid, b]).random(rows) < 0.2
            lens = np.diff(offs)
            kept = data[:n][np.repeat(~blank, lens)]
            offs[1:] = np.cumsum(np.where(blank, 0, lens))
            data[: kept.size] = kept
            n = int(kept.size)
        elif name == "Title":
            n = N.load_bench().lc_synth_title_batch(s, b, rows, min(1750, rows), 900, offs.ctypes.data, data.ctypes.data, data.size)
        elif name == "MobilePhoneModel":
            n = N.load_bench().lc_synth_phrase_batch(s, b, rows, 160, 940, offs.ctypes.data, data.ctypes.data, data.size)
        else:
            n = N.load_bench().lc_synth_phrase_batch(s, b, rows, 600, 870, offs.ctypes.data, data.ctypes.data, data.size)
        return pa.StringArray.from_buffers(rows, pa.py_buffer(offs), pa.py_buffer(data[:max(n, 1)].copy()))
    rng = np.random.default_rng([seed, cid, b])
    if name == "EventDate":       # July 2013 with a little June / August
        v = rng.integers(_days("2013-06-25"), _days("2013-08-05") + 1, size=rows).astype(np.uint16)
    elif name == "CounterID":     # Zipf-ish counters; 62 is one of the busy ones
        v = np.where(rng.random(rows) < 0.007, 62, rng.integers(1, 120_000, size=rows)).astype(np.int32)
    elif name == "UserID":
        v = rng.integers(1 << 58, 1 << 62, size=rows, dtype=np.int64)
        v[rng.random(rows) < 2e-5] = 435090932899640449
    elif name in ("RefererHash", "URLHash"):
        v = rng.integers(-(1 << 62), 1 << 62, size=rows, dtype=np.int64)
        v[rng.random(rows) < 3e-4] = 3594120000172545465 if name == "RefererHash" else 2868770270353813622
    elif name == "AdvEngineID":
        v = np.where(rng.random(rows) < 0.95, 0, rng.integers(1, 61, size=rows)).astype(np.int16)
    elif name == "TraficSourceID":
        v = rng.choice(np.array([-1, 0, 1, 2, 3, 5, 6, 7], np.int16), size=rows, p=[.12, .45, .15, .1, .05, .04, .06, .03])
    elif name == "IsRefresh":
        v = (rng.random(rows) < 0.14).astype(np.int16)
    elif name == "DontCountHits":
        v = (rng.random(rows) < 0.08).astype(np.int16)
    elif name == "IsLink":
        v = (rng.random(rows) < 0.04).astype(np.int16)
    elif name == "IsDownload":
        v = (rng.random(rows) < 0.01).astype(np.int16)
    else:
        raise KeyError(name)
    return pa.array(v, type=dtype)


def columns_of(queries=None) -> List[str]:
    names = []
    for q, conj in QUERIES.items():
        if queries is not None and q not in queries:
            continue
        for c in conj:
            for t in (c.terms if isinstance(c, AnyOf) else [c]):
                if t.column not in names:
                    names.append(t.column)
    return names


def stage_hits(cache: LiquidCache, rows: int, seed: int = 7, batch_size: int = 8192, row_group_batches: int = 54,
               file_id: int = 5, threads: int = 8, names=None, keep_arrays: bool = False):
    """Stage the synthetic columns the predicates touch.  Returns (columns: name -> pushdown.Column, entry ids, arrays)."""
    names = columns_of() if names is None else list(names)
    n_batches = (rows + batch_size - 1) // batch_size
    ids = {nm: [ParquetArrayID.new(file_id, b // row_group_batches, SCHEMA[nm][0] & 0xFFFF, b % row_group_batches)
                for b in range(n_batches)] for nm in names}
    arrays = {nm: [None] * n_batches for nm in names} if keep_arrays else None

    # string columns: one task per row group (the FSST table of a (file, row group, column) path is trained on the first
    # batch staged for it, so a row group's batches are staged in order by one thread); integer columns: striped batches
    def stage_rg(task):
        nm, rg = task
        scratch = np.zeros(batch_size * 512, np.uint8) if nm in _STRINGS else None
        for b in range(rg * row_group_batches, min((rg + 1) * row_group_batches, n_batches)):
            n = min(batch_size, rows - b * batch_size)
            arr = synth_batch(nm, seed, b, n, scratch)
            cache.insert(ids[nm][b], arr, SCHEMA[nm][2])
            if arrays is not None:
                arrays[nm][b] = arr

    n_rg = (n_batches + row_group_batches - 1) // row_group_batches
    tasks = [(nm, rg) for nm in names for rg in range(n_rg)]
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        list(ex.map(stage_rg, tasks))
    columns = {nm: Column(cache.scan(ids[nm]), SCHEMA[nm][1], SCHEMA[nm][2], fixed_width=nm not in _STRINGS) for nm in names}
    return columns, ids, arrays
