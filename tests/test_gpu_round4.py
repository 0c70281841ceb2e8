"""Round 4: the scan-level signature index (k_like_flat) against the oracle.

A scan whose entries SHARE symbol tables in runs (row groups), so that a wave's group really holds several entries:
dictionaries from one value to 2,500, row counts that are no multiple of 64, all-null entries inside a group, a group cut
by the 128-word limit and by a symbol-table change.  LIKE / NOT LIKE, with and without a selection, through k_like_flat for
every needle, through the automatic plan, and through k_like_lean — all equal to the oracle's per-entry masks and counts
(reference: byte_view_array/comparisons.rs:159-183, 598-651).
"""
import os
import sys

import numpy as np
import pyarrow as pa
import pytest

import liquid_cache_amd as lc

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_data as fz  # noqa: E402

pytestmark = pytest.mark.gpu
HINT = lc.CacheExpression.SUBSTRING_SEARCH

# (rows, distinct values, nulls) per entry; None = an all-null entry
RUNS = [
    [(8192, 2200, False), (8192, 2250, True), (8192, 2150, False), (8192, 2200, False), (10, 8, False)],
    [(65, 40, True), None, (700, 300, False), (1500, 1, False), (64, 64, False), (129, 3, True), (2049, 900, False),
     (8192, 2500, True), (8192, 2500, False), (8192, 2500, False), (8192, 2500, False), (1, 1, False)],
    [(8192, 5000, False), (8192, 4100, True), (8192, 60, False), (100, 100, False)],
]


def _want_full(lo, liquid, st, op, literal, sel, n):
    r = lo.eval_predicate(liquid, op, literal, sel, symtab=st)
    v = r.values if r.validity is None else (r.values & r.validity)
    hit = np.zeros(n, bool)
    if sel is None:
        hit[:] = v
    else:
        hit[np.flatnonzero(sel)] = v
    return hit


@pytest.fixture(scope="module")
def grouped_cases(oracle):
    lo = oracle
    rng = np.random.default_rng(77)
    runs = []
    for r_i, run in enumerate(RUNS):
        pools = []
        for e_i, spec in enumerate(run):
            if spec is None:
                pools.append(None)
                continue
            n, d, nulls = spec
            pool = fz._pool_urls(rng, d) if (e_i + r_i) % 3 else fz._pool_escape_heavy(rng, d)
            pool = list(dict.fromkeys(pool))
            while len(pool) < d:  # distinct values, exactly d of them
                pool.append(pool[int(rng.integers(len(pool)))] + b"#%d" % len(pool))
            pool = pool[:d]
            rows = [pool[i] for i in range(min(d, n))] + [pool[int(k)] for k in rng.integers(0, d, size=max(0, n - d))]
            rows = [rows[int(i)] for i in rng.permutation(len(rows))]
            if nulls:
                for i in rng.choice(n, size=max(1, n // 15), replace=False):
                    rows[int(i)] = None
            pools.append(rows)
        train = [v for rows in pools if rows for v in rows if v is not None][:20000] or [b""]
        o, dt, _ = lo.strings_to_arrow(train)
        st = lo.fsst_train(o, dt)
        entries = []
        for rows in pools:
            if rows is None:
                rows = [None] * 37
            liquid, _ = lo.encode_byte_view(rows, st=st, fingerprints=True, arrow_type=lo.BT_BINARY)
            entries.append((rows, liquid))
        runs.append((st, entries))
    return runs


@pytest.mark.parametrize("like_path", [4, 0, 3, 5])
def test_flat_index_groups_against_oracle(product_lib, oracle, grouped_cases, like_path):
    lo = oracle
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1, like_path=like_path or None).build()
    try:
        ids, flat = [], []
        for r_i, (st, entries) in enumerate(grouped_cases):
            path = 7000 + r_i
            cache.set_symbol_table(path, lo.symtab_bytes(st))
            for e_i, (rows, liquid) in enumerate(entries):
                eid = lc.ParquetArrayID.new(11, r_i, 5, e_i)
                cache.stage([eid], [liquid], [path])
                ids.append(eid)
                flat.append((rows, liquid, st))
        scan = cache.scan(ids)
        lens = [len(c[0]) for c in flat]
        offs = scan.segment_offsets
        rng = np.random.default_rng(99)
        needles = [b"google", b"mail", b"go", b"a", b"yandex.google", b"search1", b"zzzzqqq", b"http://", b"tours4",
                   b"index.php?id=1", b"\xd0\xbf\xd0\xbe", b"#21", b"gm", b"ogle.goo.gle", b"//"]
        for rows, _, st in (flat[1], flat[7], flat[-3]):
            needles += fz.make_needles(rng, rows, st, 3, for_like=True)
        n_checked = 0
        for qi, nd in enumerate(needles):
            for op, with_sel in (("like", qi % 2 == 0), ("not_like", qi % 3 == 0), ("like", qi % 2 == 1)):
                sels, words = [None] * len(flat), None
                if with_sel:
                    words = np.zeros(int(scan.mask_words), np.uint64)
                    for b, n in enumerate(lens):
                        se = rng.random(n) < [0.02, 0.5, 0.97][b % 3]
                        if b % 7 == 3:
                            se[:] = False
                        sels[b] = se
                        packed = np.packbits(se, bitorder="little")
                        words[int(offs[b]): int(offs[b + 1])].view(np.uint8)[: len(packed)] = packed
                expr = lc.LiquidExpr.try_new(op, b"%" + nd + b"%", pa.binary(), HINT)
                mask, counts = scan.eval_to_host(expr, selection=words)
                bits = np.unpackbits(mask.view(np.uint8), bitorder="little")
                for b, (rows, liquid, st) in enumerate(flat):
                    got = bits[int(offs[b]) * 64: int(offs[b]) * 64 + lens[b]].astype(bool)
                    want = _want_full(lo, liquid, st, lo.OP_NAMES[op], b"%" + nd + b"%", sels[b], lens[b])
                    assert np.array_equal(got, want), (like_path, b, op, nd, with_sel, int(got.sum()), int(want.sum()))
                    assert int(counts[b]) == int(want.sum()), (like_path, b, op, nd)
                    tail = bits[int(offs[b]) * 64 + lens[b]: int(offs[b + 1]) * 64]
                    assert not tail.any()
                    n_checked += 1
        assert n_checked >= len(flat) * 3 * len(needles)
        # the evaluation path is what was asked for
        how = scan.explain(lc.LiquidExpr.try_new("like", b"%google%", pa.binary(), HINT))
        if like_path == 4:
            assert how.startswith("k_like_flat"), how
        if like_path == 3:
            assert how.startswith("k_like_lean"), how
        if like_path == 5:
            assert how.startswith("k_like_scanall"), how
        scan.close()
    finally:
        cache.close()


# ------------------------------------------------------------------------------------------------------------------
# concurrent callers at the boundary (tests/c_abi/concurrent_callers.c: pthreads, no Python in the measured region)
# ------------------------------------------------------------------------------------------------------------------
def _build_concurrent(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "concurrent_callers")
    libdir = os.path.join(root, "liquid_cache_amd")
    subprocess.run(["gcc", "-std=gnu11", "-O1", "-Wall", "-Werror", "-pthread", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_abi", "concurrent_callers.c"), "-o", exe,
                    "-L", libdir, "-l:libliquid_cache_amd.so", "-Wl,-rpath," + libdir], check=True)
    return exe


def test_eight_threads_with_their_own_streams_while_a_ninth_stages_and_evicts(product_lib, tmp_path):
    """Eight host threads x distinct streams loop over lc_eval_predicate / lc_eval_predicate_batch / lc_scan_eval /
    lc_get_with_selection on different columns while a ninth thread stages, re-stages and evicts: every answer equals the
    single-threaded one and the eight together take less than twice one alone (no device-wide synchronise in any call;
    reference: liquid_cache_reader.rs:297-391, cache/index.rs:30-34)."""
    import subprocess
    exe = _build_concurrent(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "concurrent callers ok" in r.stdout


def test_stale_or_malformed_index_blob_is_rebuilt(product_lib, oracle):
    """an "LCIX" blob kept for ANOTHER version of an entry with the same dictionary size and row count
    must not be taken (its signatures / row lists belong to other strings), and section sizes whose sum wraps must not
    pass the header check.  Both cost a rebuild, never a result."""
    import struct
    lo = oracle
    rows_a = [b"http://alpha.example/%d" % (i % 50) for i in range(600)]
    rows_b = [r.replace(b"alpha", b"gamma") if r.endswith(b"/7") else r for r in rows_a]  # same d and n, one other string
    o, dt, _ = lo.strings_to_arrow(rows_a + rows_b)
    st = lo.fsst_train(o, dt)
    liquid_a, _ = lo.encode_byte_view(rows_a, st=st, fingerprints=True, arrow_type=lo.BT_BINARY)
    liquid_b, _ = lo.encode_byte_view(rows_b, st=st, fingerprints=True, arrow_type=lo.BT_BINARY)
    cache = lc.LiquidCacheBuilder.new().build()
    try:
        cache.set_symbol_table(8100, lo.symtab_bytes(st))
        cache.stage([1], [liquid_a], [8100])
        cache.stage([2], [liquid_b], [8100])
        blob_a, blob_b = cache.entry_index_bytes(1), cache.entry_index_bytes(2)
        assert len(blob_a) == len(blob_b) and blob_a != blob_b
        hdr = struct.unpack_from("<6I3Q", blob_a)
        assert hdr[0] == 0x5849434C and hdr[1] == 2 and hdr[8] != 0  # magic, version, content hash
        # B staged with A's blob: rebuilt (== B's own index), and the answers are B's
        cache.stage([3], [liquid_b], [8100], index_bytes=[blob_a])
        assert cache.entry_index_bytes(3) == blob_b
        for nd, rows, eid in ((b"gamma", rows_b, 3), (b"alpha.example/7", rows_b, 3), (b"alpha.example/7", rows_a, 1)):
            expr = lc.LiquidExpr.try_new("like", b"%" + nd + b"%", pa.binary(), HINT)
            got = cache.eval_predicate(eid, expr).read().to_numpy(zero_copy_only=False)
            assert [bool(g) for g in got] == [nd in r for r in rows], (nd, eid)
        # section sizes that only add up modulo 2^64: flags = row lists only, sig_bytes wraps the sum back to the length
        magic, ver, d, n, bits, flags, sig_b, post_b, h = hdr
        body = len(blob_a) - 48
        assert flags == 3 and sig_b + post_b == body
        bads = (
            # the sum of the sections wraps back to the blob's length
            struct.pack("<6I3Q", magic, ver, d, n, bits, 2, (1 << 64) - 4096, (body + 4096) % (1 << 64), h) + blob_a[48:],
            # signature bytes declared although the flag says there are none (the row lists would be read behind them)
            struct.pack("<6I3Q", magic, ver, d, n, bits, 2, sig_b, post_b, h) + blob_a[48:],
            # no content hash
            blob_a[:40] + b"\0" * 8 + blob_a[48:],
            # the previous header version
            struct.pack("<6I2Q", magic, 1, d, n, bits, flags, sig_b, post_b) + blob_a[48:])
        for bad in bads:
            cache.stage([4], [liquid_a], [8100], index_bytes=[bad])
            assert cache.entry_index_bytes(4) == blob_a
    finally:
        cache.close()


# ------------------------------------------------------------------------------------------------------------------
# the reference's own known-answer vectors (tests/golden/reference_known_answers.json, transcribed from its Rust tests with
# file:line in every record) through the HIP path — every category in one test
# ------------------------------------------------------------------------------------------------------------------
def _tri_arrow(arr):
    return [None if v is None else bool(v) for v in arr.to_pylist()]


def test_reference_known_answers_through_the_hip_path(product_lib):
    import datetime
    import json
    KA = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_known_answers.json"),
                        encoding="utf-8"))
    PA = {"int8": pa.int8(), "int16": pa.int16(), "int32": pa.int32(), "int64": pa.int64(), "uint8": pa.uint8(),
          "uint16": pa.uint16(), "uint32": pa.uint32(), "uint64": pa.uint64(), "date32": pa.date32(), "date64": pa.date64(),
          "float32": pa.float32(), "float64": pa.float64()}
    cache = lc.LiquidCacheBuilder.new().build()
    eid = [0]

    def fresh():
        eid[0] += 1
        return lc.ParquetArrayID.new(21, eid[0] >> 8, 1, eid[0] & 255)

    def typed(vals, tname):
        t = PA[tname]
        if tname == "date32":
            return pa.array(vals, pa.int32()).cast(t)
        if tname == "date64":
            return pa.array(vals, pa.int64()).cast(t)
        return pa.array(vals, t)

    n_checked = 0
    try:
        # ---- byte views: round trip + every (operator, needle) table of byte_view_array/tests.rs
        for case in KA["byte_view"]:
            values = case["values"]
            arr = pa.array(values, pa.string())
            e = fresh()
            cache.insert(e, arr, HINT if case.get("fingerprints") else None)
            assert cache.get(e).read().to_pylist() == values, case["src"]
            for op, needle, expect in case["cases"]:
                hint = HINT if op in ("like", "not_like") else None
                expr = lc.LiquidExpr.try_new(op, needle.encode(), pa.string(), hint)
                assert _tri_arrow(cache.eval_predicate(e, expr).read()) == expect, (case["src"], op, needle)
                n_checked += 1
        # ---- the len-byte-255 rule (tests.rs:755-774)
        c = KA["byte_view_long"]
        la = c["common"] + "a" * (c["long_len"] - len(c["common"]))
        lb = c["common"] + "b" * (c["long_len"] - len(c["common"]))
        e = fresh()
        cache.insert(e, pa.array([la, lb, "z"], pa.string()))
        for needle, expect in ((la, [True, False, False]), (c["common"] + "a" * 200, [False, False, False]), (lb, [False, True, False])):
            expr = lc.LiquidExpr.try_new("eq", needle.encode(), pa.string())
            assert _tri_arrow(cache.eval_predicate(e, expr).read()) == expect
            n_checked += 1
        # ---- a garbage key under a null slot is never dereferenced (tests.rs:788-806): Dictionary<u16, Utf8> input
        c = KA["byte_view_null_key_garbage"]
        keys = pa.array([k if v else None for k, v in zip(c["keys"], c["validity"])], pa.uint16())
        e = fresh()
        cache.insert(e, pa.DictionaryArray.from_arrays(keys, pa.array(c["dict"], pa.string())))
        expr = lc.LiquidExpr.try_new("eq", b"alpha", pa.string())
        assert _tri_arrow(cache.eval_predicate(e, expr).read()) == c["eq_alpha"]
        n_checked += 1
        # ---- integers: round trips incl. type extremes, all-null, single value (primitive_array.rs:771-925)
        for case in KA["primitive_roundtrip"]:
            arr = typed(case["values"], case["type"])
            e = fresh()
            cache.insert(e, arr)
            got = cache.get(e).read()
            assert got.type == arr.type and got.equals(arr), case["src"]
            n_checked += 1
        # ---- get().with_selection() (primitive_array.rs:884-982, README.md:43-60)
        for case in KA["primitive_filter"]:
            arr = typed(case["values"], case["type"])
            e = fresh()
            cache.insert(e, arr)
            got = cache.get(e).with_selection(np.array(case["selection"], bool)).read()
            assert got.to_pylist() == case["expect"], case["src"]
            n_checked += 1
        # ---- README predicate
        for case in KA["primitive_predicate"]:
            arr = typed(case["values"], case["type"])
            e = fresh()
            cache.insert(e, arr)
            expr = lc.LiquidExpr.try_new(case["op"], case["literal"], arr.type)
            assert _tri_arrow(cache.eval_predicate(e, expr).read()) == case["expect"]
            n_checked += 1
        # ---- serialized size of Date32 0..4096 (cache/tests/snapshots: 6,184 bytes)
        for case in KA["serialized_size"]:
            arr = typed(list(range(*case["range"])), case["type"])
            assert len(cache.transcode(arr)) == case["bytes"]
            e = fresh()
            cache.insert(e, arr)
            assert cache.get(e).read().equals(arr)
            n_checked += 1
        # ---- floats (float_array.rs:1082-1122)
        for case in KA["float_roundtrip"]:
            arr = typed(case["values"], case["type"])
            e = fresh()
            cache.insert(e, arr)
            got = cache.get(e).read()
            assert got.type == arr.type and got.to_pylist() == arr.to_pylist(), case["src"]
            n_checked += 1
        # ---- bit packing at the widths of bit_pack_array.rs:356-531 (through the cache: values (i mod 2^W))
        for case in KA["bit_pack"]:
            vals = [v % (1 << case["bit_width"]) for v in range(*case["range"])]
            arr = typed(vals, case["type"])
            e = fresh()
            cache.insert(e, arr)
            assert cache.get(e).read().equals(arr), case
            lit = vals[len(vals) // 2]
            expr = lc.LiquidExpr.try_new("ge", lit, arr.type)
            assert cache.eval_predicate(e, expr).read().to_pylist() == [v >= lit for v in vals]
            n_checked += 1
        # ---- boolean_buffer_and_then (datafusion/src/utils.rs:54-57, :316-408)
        for case in KA["and_then"]:
            b = lambda s: np.array([ch == "Y" for ch in s], bool)  # noqa: E731
            got = lc.boolean_buffer_and_then(cache, b(case["left"]), b(case["right"]))
            assert "".join("Y" if x else "N" for x in got) == case["expect"], case["src"]
            n_checked += 1
        # ---- date parts (squeezed_date32_array.rs:520-618): extraction and the lossy reconstruction
        fields = {"year": lc.Date32Field.YEAR, "month": lc.Date32Field.MONTH, "day": lc.Date32Field.DAY,
                  "dow": lc.Date32Field.DAY_OF_WEEK}
        iso = lambda d: None if d is None else datetime.date.fromisoformat(d)  # noqa: E731
        for field, dates, expect in KA["date_parts"]["lossy"]:
            arr = pa.array([iso(d) for d in dates], pa.date32())
            e = fresh()
            cache.insert(e, arr)
            got = cache.get(e).with_expression_hint(lc.CacheExpression.extract_date32(fields[field])).read()
            assert got.to_pylist() == [iso(d) for d in expect], (field, dates)
            n_checked += 1
        # ---- reader level (liquid_cache_reader.rs:783-892): predicate, and_then, get().with_selection()
        for case in KA["reader_level"]:
            out = []
            for batch in case["batches"]:
                arr = pa.array(batch, pa.int32())
                e = fresh()
                cache.insert(e, arr)
                sel = np.array(case.get("selection", [True] * len(batch)), bool)
                expr = lc.LiquidExpr.try_new(case["op"], case["literal"], pa.int32())
                r = cache.eval_predicate(e, expr).with_selection(sel).read()
                mask = np.array([bool(v) for v in r.fill_null(False).to_pylist()], bool)
                final = lc.boolean_buffer_and_then(cache, sel, mask)
                if final.any():  # (an empty selection skips the batch: liquid_cache_reader.rs:309-311)
                    out += cache.get(e).with_selection(final).read().to_pylist()
            assert out == case["expect_rows"], case["src"]
            n_checked += 1
    finally:
        cache.close()
    assert n_checked >= 85


def test_scan_level_index_survives_the_scan(product_lib, oracle, grouped_cases):
    """A host that creates a scan per query over the same entries (DataFusion builds a reader per query) gets the scan-level
    index and the plans of the previous scan: same results, no rebuild; a re-staged entry (new publication) is not served
    from the old index."""
    import time
    from liquid_cache_amd import _native as N
    lo = oracle
    # (LC_OPT_SCAN_CACHE = 0: this test is about the INDEX cache behind really destroyed scans; the scan cache of round 6, which
    # would hand the whole scan back, has its own test in test_gpu_round6.py)
    # ... and LC_OPT_LIKE_INDEX_ASYNC = 0: the plan text compared below is the one made ON the scan-level index
    cache = (lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1).with_option(N.OPT_SCAN_CACHE, 0)
             .with_option(N.OPT_LIKE_INDEX_ASYNC, 0).build())
    sync_builds = True
    try:
        ids, flat = [], []
        for r_i, (st, entries) in enumerate(grouped_cases):
            path = 7100 + r_i
            cache.set_symbol_table(path, lo.symtab_bytes(st))
            for e_i, (rows, liquid) in enumerate(entries):
                eid = lc.ParquetArrayID.new(12, r_i, 5, e_i)
                cache.stage([eid], [liquid], [path])
                ids.append(eid)
                flat.append((rows, liquid, st, path))
        expr = lc.LiquidExpr.try_new("like", b"%google%", pa.binary(), HINT)
        one = lc.LiquidExpr.try_new("like", b"%a%", pa.binary(), HINT)

        def run(scan):
            t0 = time.perf_counter()
            m, c = scan.eval_to_host(expr)
            dt = time.perf_counter() - t0
            m1, c1 = scan.eval_to_host(one)
            return m.copy(), c.copy(), m1.copy(), c1.copy(), dt

        s1 = cache.scan(ids)
        a = run(s1)
        how1 = s1.explain(expr)
        assert "scan-level index" in how1, how1
        s1.close()
        s2 = cache.scan(ids)
        how2 = s2.explain(expr)          # before any evaluation on THIS scan: nothing adopted yet
        assert how2.startswith("k_str_pred (scan not evaluated yet)"), how2
        b = run(s2)
        how2 = s2.explain(expr)
        s2.close()
        assert how2 == how1, (how1, how2)  # the same plan and the same index (its build time is part of the text)
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y)
        # (the index build shows in the first evaluation of a scan that has to make it — when that evaluation waits for it)
        if sync_builds:
            assert b[4] < a[4], (a[4], b[4])
        # re-stage one entry with other rows: the scan over the new publication answers from ITS data
        rows, liquid, st, path = flat[2]
        other = [None if v is None else v + b"~google~" for v in rows]
        liquid2, _ = lo.encode_byte_view(other, st=st, fingerprints=True, arrow_type=lo.BT_BINARY)
        cache.stage([ids[2]], [liquid2], [path])
        s3 = cache.scan(ids)
        m3, c3 = s3.eval_to_host(expr)
        offs = s3.segment_offsets
        bits = np.unpackbits(m3.view(np.uint8), bitorder="little")
        got = bits[int(offs[2]) * 64: int(offs[2]) * 64 + len(other)].astype(bool)
        assert np.array_equal(got, np.array([v is not None for v in other]))
        assert int(c3[2]) == sum(v is not None for v in other)
        for b_i in (0, 1, 3):
            assert int(c3[b_i]) == int(a[1][b_i])
        s3.close()
    finally:
        cache.close()
