"""Round 5: sparse results — hit lists instead of masks, and get-with-selection from a hit list in one launch.

Reference shape: the per-batch BooleanBuffer a filter leaves and the gathers that consume it
(datafusion/src/reader/runtime/liquid_cache_reader.rs:297-391, core/src/liquid_array/byte_view_array/helpers.rs:44-64,
primitive_array.rs:370-374).  A hit list must hold exactly the set bits of the mask the oracle computes; the values gathered
for its records must be the oracle's get-with-selection of the same rows.
"""
import os
import sys

import numpy as np
import pyarrow as pa
import pytest

import liquid_cache_amd as lc

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_data as fz  # noqa: E402
from test_gpu_round4 import RUNS, _want_full, grouped_cases  # noqa: E402,F401  (the grouped-entries fixture)

pytestmark = pytest.mark.gpu
HINT = lc.CacheExpression.SUBSTRING_SEARCH


def _stage_grouped(cache, lo, cases):
    ids, flat = [], []
    for r_i, (st, entries) in enumerate(cases):
        path = 7100 + r_i
        cache.set_symbol_table(path, lo.symtab_bytes(st))
        for e_i, (rows, liquid) in enumerate(entries):
            eid = lc.ParquetArrayID.new(12, r_i, 5, e_i)
            cache.stage([eid], [liquid], [path])
            ids.append(eid)
            flat.append((rows, liquid, st))
    return ids, flat


def _hits_to_bits(hits, offs, n_bits):
    bits = np.zeros(n_bits, bool)
    e = (hits >> np.uint64(32)).astype(np.int64)
    r = (hits & np.uint64(0xFFFFFFFF)).astype(np.int64)
    pos = offs[e].astype(np.int64) * 64 + r
    assert len(np.unique(pos)) == len(pos), "a row appears twice in the hit list"
    bits[pos] = True
    return bits


def _check_list_shape(hits, counts, first):
    """The records of one entry are contiguous and ascending; `first` points at each entry's first record."""
    if len(hits) == 0:
        return
    e = (hits >> np.uint64(32)).astype(np.int64)
    r = (hits & np.uint64(0xFFFFFFFF)).astype(np.int64)
    change = np.flatnonzero(np.diff(e) != 0) + 1
    starts = np.concatenate([[0], change])
    ents = e[starts]
    assert len(np.unique(ents)) == len(ents), "the records of an entry are not contiguous"
    same = np.diff(e) == 0
    assert (np.diff(r)[same] > 0).all(), "rows of an entry are not ascending"
    for s, en in zip(starts, ents):
        assert int(first[en]) == int(s)
    run_len = np.diff(np.concatenate([starts, [len(hits)]]))
    assert run_len.tolist() == counts[ents].tolist()


@pytest.mark.parametrize("like_path", [0, 4, 3, 1])
def test_hit_list_equals_oracle_mask(product_lib, oracle, grouped_cases, like_path):
    lo = oracle
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1, like_path=like_path or None).build()
    try:
        ids, flat = _stage_grouped(cache, lo, grouped_cases)
        scan = cache.scan(ids)
        lens = [len(c[0]) for c in flat]
        offs = scan.segment_offsets
        n_bits = int(scan.mask_words) * 64
        rng = np.random.default_rng(5)
        needles = [b"google", b"mail", b"go", b"a", b"zzzzqqq", b"http://", b"index.php?id=1", b"#21", b"//"]
        for rows, _, st in (flat[1], flat[7]):
            needles += fz.make_needles(rng, rows, st, 2, for_like=True)
        checked = 0
        for qi, nd in enumerate(needles):
            for op, with_sel in (("like", qi % 2 == 0), ("not_like", qi % 3 == 0), ("like", qi % 2 == 1)):
                sels, words = [None] * len(flat), None
                if with_sel:
                    words = np.zeros(int(scan.mask_words), np.uint64)
                    for b, n in enumerate(lens):
                        se = rng.random(n) < [0.02, 0.5, 0.97][b % 3]
                        sels[b] = se
                        packed = np.packbits(se, bitorder="little")
                        words[int(offs[b]): int(offs[b + 1])].view(np.uint8)[: len(packed)] = packed
                expr = lc.LiquidExpr.try_new(op, b"%" + nd + b"%", pa.binary(), HINT)
                want = np.zeros(n_bits, bool)
                want_counts = np.zeros(len(flat), np.int64)
                for b, (rows, liquid, st) in enumerate(flat):
                    w = _want_full(lo, liquid, st, lo.OP_NAMES[op], b"%" + nd + b"%", sels[b], lens[b])
                    want[int(offs[b]) * 64: int(offs[b]) * 64 + lens[b]] = w
                    want_counts[b] = int(w.sum())
                for from_mask in (False, True):
                    hits, n, counts, total, first = scan.eval_hits_to_host(expr, selection=words, from_mask=from_mask)
                    assert n == len(hits) == int(want.sum()) == total, (nd, op, with_sel, from_mask, n, int(want.sum()), total)
                    assert counts.astype(np.int64).tolist() == want_counts.tolist()
                    assert np.array_equal(_hits_to_bits(hits, offs, n_bits), want), (nd, op, with_sel, from_mask)
                    _check_list_shape(hits, counts, first)
                # a buffer that is too small: the count is still the full one, what was written are records of the result
                if int(want.sum()) > 3:
                    cap = int(want.sum()) // 2
                    hits, n, _, total, _ = scan.eval_hits_to_host(expr, selection=words, capacity=cap)
                    assert n == total == int(want.sum()) and len(hits) == cap
                    assert _hits_to_bits(hits, offs, n_bits)[~want].sum() == 0
                checked += 1
        assert checked >= 27
        scan.close()
    finally:
        cache.close()


def test_count_without_mask_and_string_equality_hits(product_lib, oracle, grouped_cases):
    """d_mask_out = NULL is legal for COUNT(*) / per-entry-count consumers on every evaluation path; string = / <> through
    the scan-level index emits hit lists as well."""
    import ctypes as C
    lo = oracle
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1).build()
    try:
        ids, flat = _stage_grouped(cache, lo, grouped_cases)
        scan = cache.scan(ids)
        offs = scan.segment_offsets
        n_bits = int(scan.mask_words) * 64
        lens = [len(c[0]) for c in flat]
        some_value = next(v for v in flat[0][0] if v is not None and len(v) > 14)
        exprs = [lc.LiquidExpr.try_new("like", b"%google%", pa.binary(), HINT),
                 lc.LiquidExpr.try_new("not_like", b"%a%", pa.binary(), HINT),
                 lc.LiquidExpr.try_new("=", some_value, pa.binary(), HINT),
                 lc.LiquidExpr.try_new("!=", some_value, pa.binary(), HINT),
                 lc.LiquidExpr.try_new("<", b"http://m", pa.binary(), HINT)]
        for expr in exprs:
            mask, counts = scan.eval_to_host(expr)
            d_total = scan._to_dev(np.zeros(1, np.uint64))
            d_counts = scan._dev(max(scan.entries, 1) * 4)
            try:
                scan.eval_count(expr, 0, d_total.value, 0, d_counts.value)
                cache._lib.lc_stream_synchronize(cache.handle, None)
                total = int(scan._from_dev(d_total, np.uint64, 1)[0])
                c2 = scan._from_dev(d_counts, np.uint32, scan.entries)
            finally:
                cache._lib.lc_device_free(cache.handle, d_total)
                cache._lib.lc_device_free(cache.handle, d_counts)
            assert total == int(counts.sum()) and c2.tolist() == counts.tolist()
            hits, n, c3, t3, first = scan.eval_hits_to_host(expr)
            bits = np.unpackbits(mask.view(np.uint8), bitorder="little").astype(bool)
            assert n == t3 == total and np.array_equal(_hits_to_bits(hits, offs, n_bits), bits)
            _check_list_shape(hits, c3, first)
        # no output at all is an error, not a silent no-op
        pred = exprs[0].as_predicate()
        st = cache._lib.lc_scan_eval(cache.handle, scan._h, C.byref(pred), None, None, None, None)
        assert st == -1
        assert sum(lens) == scan.rows
        scan.close()
    finally:
        cache.close()


def test_gather_bytes_from_hit_list(product_lib, oracle, grouped_cases):
    """BinaryView records + data buffer for the rows of a hit list == the oracle's get-with-selection of those rows."""
    lo = oracle
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1).build()
    try:
        ids, flat = _stage_grouped(cache, lo, grouped_cases)
        scan = cache.scan(ids)
        lens = [len(c[0]) for c in flat]
        truth = []  # per entry: the oracle's decode of every row
        for rows, liquid, st in flat:
            truth.append(lo.filter_byte_view(liquid, st, None))
            assert truth[-1] == list(rows)
        rng = np.random.default_rng(8)
        # (a) what a selective LIKE leaves: the wave-cooperative decode (few rows per wave)
        for nd in (b"google", b"#21", b"zzzzqqq"):
            expr = lc.LiquidExpr.try_new("like", b"%" + nd + b"%", pa.binary(), HINT)
            hits, n, _, _, _ = scan.eval_hits_to_host(expr)
            got = scan.gather_bytes_hits_to_host(hits)
            want = [truth[int(h >> np.uint64(32))][int(h & np.uint64(0xFFFFFFFF))] for h in hits]
            assert got == want and all(nd in v for v in got)
            assert scan.gather_bytes_hits_to_host(hits, slotted=True) == want   # LC_GATHER_SLOTTED: same array, sparse buffer
        # (b) random rows incl. nulls, in list order (any order is legal input): small and large lists (lane-per-row decode)
        all_refs = np.concatenate([(np.uint64(e) << np.uint64(32)) | np.arange(n, dtype=np.uint64) for e, n in enumerate(lens)])
        for k in (1, 7, 63, 64, 65, 900, len(all_refs)):
            refs = all_refs if k == len(all_refs) else all_refs[rng.choice(len(all_refs), size=k, replace=False)]
            got = scan.gather_bytes_hits_to_host(refs, capacity_bytes=64)  # (too small at first: retried with *d_n_bytes)
            want = [truth[int(h >> np.uint64(32))][int(h & np.uint64(0xFFFFFFFF))] for h in refs]
            assert got == want, k
            # (slotted: the slots always fit, the values longer than a slot are what the retry is for)
            assert scan.gather_bytes_hits_to_host(refs, capacity_bytes=64, slotted=True) == want, k
        scan.close()
    finally:
        cache.close()


def test_gather_bytes_hits_length_classes(gpu_cache, oracle):
    """Values of 0, 1..12 (inline views), 13, 254, 255 (length byte saturates) and 700 bytes, escapes, a shared prefix."""
    rng = np.random.default_rng(3)
    lengths = [0, 1, 4, 11, 12, 13, 14, 100, 104, 105, 106, 127, 128, 129, 253, 254, 255, 256, 700]
    pool = [bytes(rng.integers(32, 127, size=n, dtype=np.uint8)) for n in lengths] + [b"\xff\xfe\x00\xff" * 5, "ÿñ".encode() * 9]
    for shared in (b"", b"http://www.example.com/"):
        vals = [shared + p for p in pool]
        rows = [vals[int(i)] for i in rng.integers(0, len(vals), size=3000)]
        for i in rng.choice(3000, size=100, replace=False):
            rows[int(i)] = None
        ids = []
        for b in range(3):
            eid = lc.ParquetArrayID.new(40 + len(shared), 0, 9, b)
            gpu_cache.insert(eid, pa.array(rows[b * 1000:(b + 1) * 1000], type=pa.binary()), HINT)
            ids.append(eid)
        scan = gpu_cache.scan(ids)
        refs = np.concatenate([(np.uint64(e) << np.uint64(32)) | np.arange(1000, dtype=np.uint64) for e in range(3)])
        for sub in (refs, refs[rng.permutation(len(refs))[:50]]):
            got = scan.gather_bytes_hits_to_host(sub)
            want = [rows[int(h >> np.uint64(32)) * 1000 + int(h & np.uint64(0xFFFFFFFF))] for h in sub]
            assert got == want
            # values of exactly 128 bytes and of 129 sit on the two sides of the slot size (shared prefix 23 + 105 / 106 ...)
            assert scan.gather_bytes_hits_to_host(sub, slotted=True) == want
        scan.close()


def test_gather_fixed_from_hit_list(gpu_cache, oracle):
    """Fixed-width columns: integers of several lane types, dates, decimals, ALP floats with patches; hits of a predicate on
    ONE column project ANOTHER column of the same row ranges."""
    import decimal
    rng = np.random.default_rng(12)
    n_batches, n = 5, 8192
    total = n_batches * n - 100
    cols = {
        "i64": (rng.integers(-2**40, 2**40, size=total).astype(np.int64), pa.int64(), np.int64),
        "i16": (rng.integers(-300, 3000, size=total).astype(np.int16), pa.int16(), np.int16),
        "u8": (rng.integers(0, 200, size=total).astype(np.uint8), pa.uint8(), np.uint8),
        "d32": (rng.integers(7000, 12000, size=total).astype(np.int32), pa.date32(), np.int32),
        "f64": ((rng.integers(-10**6, 10**6, size=total) / 100.0).astype(np.float64), pa.float64(), np.float64),
        "f32": ((rng.integers(-10**4, 10**4, size=total) / 10.0).astype(np.float32), pa.float32(), np.float32),
    }
    cols["f64"][0][rng.random(total) < 0.01] = np.pi  # ALP exceptions
    cols["f32"][0][rng.random(total) < 0.01] = np.float32(np.e)
    null_mask = rng.random(total) < 0.05
    scans = {}
    for ci, (name, (vals, pa_dt, _)) in enumerate(cols.items()):
        ids = []
        for k in range(n_batches):
            eid = lc.ParquetArrayID.new(6, 0, 20 + ci, k)
            sl = slice(k * n, min((k + 1) * n, total))
            gpu_cache.insert(eid, pa.array(vals[sl], type=pa_dt, mask=null_mask[sl] if name in ("i16", "f64") else None))
            ids.append(eid)
        scans[name] = gpu_cache.scan(ids)
    dec = [decimal.Decimal(int(x)) / 100 for x in rng.integers(0, 10**7, size=total)]
    ids = []
    for k in range(n_batches):
        eid = lc.ParquetArrayID.new(6, 0, 40, k)
        gpu_cache.insert(eid, pa.array(dec[k * n: min((k + 1) * n, total)], type=pa.decimal128(15, 2)))
        ids.append(eid)
    scans["dec"] = gpu_cache.scan(ids)
    # the filter: i64 > literal (fixed-width evaluation: the list comes from the mask in scan-owned scratch)
    lit = int(np.quantile(cols["i64"][0], 0.99))
    expr = lc.LiquidExpr.try_new(">", lit, pa.int64())
    hits, n_hits, counts, tot, first = scans["i64"].eval_hits_to_host(expr)
    keep = cols["i64"][0] > lit
    assert n_hits == tot == int(keep.sum()) == len(hits)
    _check_list_shape(hits, counts, first)
    rows = (hits >> np.uint64(32)).astype(np.int64) * n + (hits & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert sorted(rows.tolist()) == np.flatnonzero(keep).tolist()
    for name, (vals, _, np_dt) in cols.items():
        got, valid = scans[name].gather_fixed_hits_to_host(hits, np_dt)
        nulls = null_mask[rows] if name in ("i16", "f64") else np.zeros(len(rows), bool)
        assert valid.tolist() == (~nulls).tolist()
        assert got[~nulls].view(np.uint8).tobytes() == vals[rows][~nulls].view(np.uint8).tobytes(), name  # bit-exact
    got, valid = scans["dec"].gather_fixed_hits_to_host(hits, np.dtype([("lo", np.uint64), ("hi", np.uint64)]))
    assert valid.all() and got["hi"].tolist() == [0] * len(rows)
    assert got["lo"].tolist() == [int(dec[r] * 100) for r in rows]
    # a dense list (every row): same values as the mask-driven gather
    all_refs = np.concatenate([(np.uint64(e) << np.uint64(32)) | np.arange(min(n, total - e * n), dtype=np.uint64)
                               for e in range(n_batches)])
    got, _ = scans["u8"].gather_fixed_hits_to_host(all_refs, np.uint8)
    assert got.tobytes() == cols["u8"][0].tobytes()
    for s in scans.values():
        s.close()


def test_cast_wrapped_predicates_on_reference_samples(gpu_cache):
    """`CAST(col) OP literal` / `to_timestamp_seconds(col) OP literal` (liquid_expr.rs:150-174): the rewritten predicate on the
    device == pyarrow's cast + compare over the reference's own sample columns (ClickBench's EventTime is an Int64 of seconds,
    EventDate a UInt16 of days — the queries wrap both)."""
    import datetime
    import pyarrow.compute as pc
    import pyarrow.parquet as pq
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    hits = pq.read_table(os.path.join(gold, "nano_hits_cols.parquet"))
    line = pq.read_table(os.path.join(gold, "lineitem_sf0001.parquet"))
    fns = {"=": pc.equal, "!=": pc.not_equal, "<": pc.less, "<=": pc.less_equal, ">": pc.greater, ">=": pc.greater_equal}
    col = lc.Column()
    et = hits["EventTime"].combine_chunks()
    t_mid = datetime.datetime.utcfromtimestamp(int(np.median(et.to_numpy())))
    cases = [
        (hits["EventTime"], lc.ToTimestampSeconds(col), lambda a: pc.cast(a, pa.timestamp("s")), pa.timestamp("s"),
         [t_mid, t_mid + datetime.timedelta(seconds=1), datetime.datetime(2013, 7, 15), datetime.datetime(1970, 1, 1)]),
        (hits["EventDate"], lc.Cast(col, pa.int32()), lambda a: pc.cast(a, pa.int32()), pa.int32(), [-5, 0, 15901, 15902, 70000]),
        (hits["ResolutionWidth"], lc.Cast(col, pa.float64()), lambda a: pc.cast(a, pa.float64()), pa.float64(),
         [1024.0, 1023.5, 1366.25, -1.0, 1e9]),
        (hits["RegionID"], lc.Cast(col, pa.int64()), lambda a: pc.cast(a, pa.int64()), pa.int64(), [229, 2**35, -2**35]),
        (line["l_shipdate"], lc.Cast(col, pa.timestamp("us")), lambda a: pc.cast(a, pa.timestamp("us")), pa.timestamp("us"),
         [datetime.datetime(1994, 1, 1), datetime.datetime(1994, 1, 1, 0, 0, 0, 1), datetime.datetime(1995, 6, 17, 12)]),
        (hits["URL"], lc.Cast(col, pa.string_view()), lambda a: a, pa.string(), ["", "http://"]),
    ]
    eid = 9000
    checked = 0
    for chunked, lhs, cast, t_lit, lits in cases:
        arr = chunked.combine_chunks()
        ids = []
        for b in range(0, len(arr), 8192):
            eid += 1
            gpu_cache.insert(eid, arr.slice(b, 8192))
            ids.append(eid)
        scan = gpu_cache.scan(ids)
        casted = cast(arr)
        for op, fn in fns.items():
            for lit in lits:
                expr = lc.LiquidExpr.try_new(op, lit, arr.type, None, lhs)
                assert expr is not None, (str(arr.type), op, lit)
                want = fn(casted, pa.scalar(lit, type=t_lit)).fill_null(False).to_numpy(zero_copy_only=False)
                mask, counts = scan.eval_to_host(expr)
                got = np.zeros(len(arr), bool)
                bits = np.unpackbits(mask.view(np.uint8), bitorder="little").astype(bool)
                for k, b in enumerate(range(0, len(arr), 8192)):
                    n = min(8192, len(arr) - b)
                    got[b:b + n] = bits[int(scan.segment_offsets[k]) * 64: int(scan.segment_offsets[k]) * 64 + n]
                assert np.array_equal(got, want), (str(arr.type), op, lit, int(got.sum()), int(want.sum()))
                assert int(counts.sum()) == int(want.sum())
                checked += 1
        scan.close()
    assert checked >= 100


def test_index_budget_and_eviction_mid_stream(product_lib, oracle, grouped_cases):
    """The scan-level LIKE index is HBM the caller's budget covers (lc_scan_info_get, max_hbm_bytes): a scan whose index does
    not fit evaluates with the entry-level index (k_like_lean) — same masks — and an index cached for a future scan is
    dropped when a live query needs the room.  A query stream over three 'columns' under a budget of 1.5 indexes."""
    lo = oracle

    def stage_cols(cache):
        cols = []
        for c in range(3):
            ids = []
            for r_i, (st, entries) in enumerate(grouped_cases):
                path = 7300 + 10 * c + r_i
                cache.set_symbol_table(path, lo.symtab_bytes(st))
                for e_i, (rows, liquid) in enumerate(entries):
                    eid = lc.ParquetArrayID.new(20 + c, r_i, 5, e_i)
                    cache.stage([eid], [liquid], [path])
                    ids.append(eid)
            cols.append(ids)
        return cols

    needles = [b"google", b"index.php?id=1", b"zzzzqqq"]
    exprs = [lc.LiquidExpr.try_new("like", b"%" + nd + b"%", pa.binary(), HINT) for nd in needles]
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1).build()
    try:
        cols = stage_cols(cache)
        scan = cache.scan(cols[0])
        want = [scan.eval_to_host(e) for e in exprs]
        info = scan.info()
        index_bytes, slab_bytes = int(info.index_bytes), int(info.ctx_slab_bytes)
        assert index_bytes > 0 and int(info.ctx_index_bytes) == index_bytes and info.like_plans == len(needles)
        assert scan.explain(exprs[2]).startswith("k_like_flat")  # (a needle the plan finds selective; 'google' is not, here)
        assert int(info.entries) == scan.entries and int(info.rows) == scan.rows and info.is_byte_view == 1
        scan.close()
    finally:
        cache.close()
    budget = slab_bytes + index_bytes * 3 // 2
    # (LC_OPT_LIKE_INDEX_ASYNC = 0: a query that needs room takes it at once here; the asynchronous policy — a scan must prove
    # hot before its build evicts another scan's index — is what test_gpu_round6.py checks)
    from liquid_cache_amd import _native as N
    cache = (lc.LiquidCacheBuilder.new().with_max_memory_bytes(budget).with_index_options(like_pipeline_min_entries=1)
             .with_option(N.OPT_LIKE_INDEX_ASYNC, 0).build())
    try:
        cols = stage_cols(cache)

        def query(c, keep=False):
            s = cache.scan(cols[c])
            for e, (m, cnt) in zip(exprs, want):
                gm, gc = s.eval_to_host(e)
                assert np.array_equal(gm, m) and gc.tolist() == cnt.tolist()
            inf = s.info()
            assert int(inf.ctx_index_bytes) + int(inf.ctx_slab_bytes) <= budget
            path = s.explain(exprs[2]).split(":")[0].split(" ")[0]
            if not keep:
                s.close()
            return path, int(inf.index_bytes), s
        seen = []
        for q in range(6):  # a scan per query, the columns round robin: every query finds room (cached indexes are dropped)
            path, ib, _ = query(q % 3)
            seen.append(path)
            assert path == "k_like_flat" and ib == index_bytes
        # a LIVE scan holds its index: the next column's index does not fit the budget and the entry-level index serves
        path_a, ib_a, live = query(0, keep=True)
        assert path_a == "k_like_flat" and ib_a == index_bytes
        path_b, ib_b, _ = query(1)
        assert path_b == "k_like_lean" and ib_b == 0, (path_b, ib_b)
        live.close()
        path_c, ib_c, _ = query(1)  # the room is back
        assert path_c == "k_like_flat" and ib_c == index_bytes
        # staging more data than the budget leaves fails like the reference's CacheFull, the cached index going first
        extra = lc.ParquetArrayID.new(29, 0, 5, 0)
        cache.stage([extra], [grouped_cases[0][1][0][1]], [7300])
    finally:
        cache.close()


def test_filter_hits_against_oracle(product_lib, oracle, grouped_cases, gpu_cache):
    """lc_scan_filter_hits: the survivors of a hit list under a predicate on the same row ranges == the rows the oracle's
    eval_predicate marks among the listed ones — byte views (compare ops, LIKE / NOT LIKE, boolean literal) and fixed width
    (integers, decimals, ALP floats with patches, nulls)."""
    lo = oracle
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1).build()
    try:
        ids, flat = _stage_grouped(cache, lo, grouped_cases)
        scan = cache.scan(ids)
        lens = [len(c[0]) for c in flat]
        rng = np.random.default_rng(21)
        all_refs = np.concatenate([(np.uint64(e) << np.uint64(32)) | np.arange(n, dtype=np.uint64) for e, n in enumerate(lens)])
        some_value = next(v for v in flat[3][0] if v is not None and len(v) > 70 or v is not None and len(v) > 20)
        preds = [("=", some_value), ("!=", some_value), ("<", b"http://m"), (">=", b"http://m"), ("<=", b""), (">", b""),
                 ("like", b"%google%"), ("not_like", b"%a%"), ("like", b"%index.php?id=1%"), ("=", b"x" * 100), ("like", b"%" + b"ab" * 40 + b"%")]
        n_checked = 0
        for op, lit in preds:
            expr = lc.LiquidExpr.try_new(op, lit, pa.binary(), HINT)
            truth = {}
            for b, (rows, liquid, st) in enumerate(flat):
                truth[b] = _want_full(lo, liquid, st, lo.OP_NAMES["==" if op == "=" else op], lit, None, lens[b])
            for k in (1, 63, 65, 700, len(all_refs)):
                refs = all_refs if k == len(all_refs) else all_refs[rng.choice(len(all_refs), size=k, replace=False)]
                got = scan.filter_hits_to_host(expr, refs)
                want = [int(h) for h in refs if truth[int(h >> np.uint64(32))][int(h & np.uint64(0xFFFFFFFF))]]
                assert sorted(int(x) for x in got) == sorted(want), (op, lit[:20], k, len(got), len(want))
                # relative order inside every 64-record batch of the input is kept
                pos = {int(h): i for i, h in enumerate(refs)}
                gi = [pos[int(x)] for x in got]
                for a, b2 in zip(gi, gi[1:]):
                    assert a // 64 != b2 // 64 or a < b2
                n_checked += 1
        assert n_checked >= 50
        # the boolean literal on byte-like columns (liquid_expr.rs:78-80): every valid listed row, or none
        for val in (True, False):
            expr = lc.LiquidExpr.try_new(None, val, pa.binary())
            got = scan.filter_hits_to_host(expr, all_refs)
            valid = [int(h) for h in all_refs if flat[int(h >> np.uint64(32))][0][int(h & np.uint64(0xFFFFFFFF))] is not None]
            assert sorted(int(x) for x in got) == (sorted(valid) if val else [])
        scan.close()
    finally:
        cache.close()
    # fixed width: the same listed rows through the oracle's predicate on the same Liquid bytes
    import decimal
    n = 8192
    arrays = [pa.array(rng.integers(-5000, 5000, size=n), type=pa.int64(), mask=rng.random(n) < 0.1),
              pa.array(rng.integers(0, 60000, size=n).astype(np.uint16)),
              pa.array(rng.integers(-120, 120, size=n).astype(np.int8)),
              pa.array(rng.integers(7000, 12000, size=n).astype(np.int32), type=pa.date32()),
              pa.array([decimal.Decimal(int(x)) / 100 for x in rng.integers(0, 10**6, size=n)], type=pa.decimal128(15, 2)),
              pa.array(np.where(rng.random(n) < 0.02, np.pi, (rng.integers(-10**5, 10**5, size=n) / 100.0)), type=pa.float64(),
                       mask=rng.random(n) < 0.05),
              pa.array(np.where(rng.random(n) < 0.02, np.float32(np.e), (rng.integers(-10**4, 10**4, size=n) / 10.0).astype(np.float32)).astype(np.float32))]
    lits = [[-4999, 0, 12, 5000, 10**12, -10**12], [0, 30000, 59999, 70000], [-120, 0, 119, 127, -200],
            [__import__("datetime").date(1992, 1, 2), __import__("datetime").date(1999, 5, 5)],
            [decimal.Decimal("5000.00"), decimal.Decimal("0.00"), decimal.Decimal("-1.00")],
            [0.0, 3.141592653589793, 12.34, -1e9, float("nan")], [0.0, float(np.float32(np.e)), 12.5]]
    refs_all = np.arange(n, dtype=np.uint64)
    eid = 8800
    n_fixed = 0
    for arr, ls in zip(arrays, lits):
        eid += 1
        liquid = gpu_cache.transcode(arr)
        gpu_cache.stage([eid], [liquid], data_types=[arr.type])
        scan = gpu_cache.scan([eid])
        for op in ("=", "!=", "<", "<=", ">", ">="):
            for lit in ls:
                expr = lc.LiquidExpr.try_new(op, lit, arr.type)
                assert expr is not None
                olit = lit
                if isinstance(lit, decimal.Decimal):
                    olit = int(lit * 100)
                elif hasattr(lit, "toordinal"):
                    olit = (lit - __import__("datetime").date(1970, 1, 1)).days
                r = lo.eval_predicate(liquid, lo.OP_NAMES["==" if op == "=" else op], olit)
                hit = r.values if r.validity is None else (r.values & r.validity)
                for refs in (refs_all, refs_all[rng.choice(n, size=300, replace=False)]):
                    got = scan.filter_hits_to_host(expr, refs)
                    want = [int(h) for h in refs if hit[int(h)]]
                    assert sorted(int(x) for x in got) == sorted(want), (str(arr.type), op, lit, len(got), len(want))
                    n_fixed += 1
        scan.close()
    assert n_fixed >= 200


def test_bench_two_rank_dry_run_on_one_gpu():
    """bench.py's multi-rank path (launcher contract, strong-scaling split by contiguous row ranges, COUNT(*) exchange, compact
    record) on ONE GPU: two ranks share device 0 and exchange through gloo (LC_BENCH_TEST_BACKEND) — logic, not a measurement.
    The union of the two shards is the table a single rank stages: same COUNT(*)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--rows", "3000000", "--steps", "2", "--warmup", "1", "--scans-per-step", "4", "--rotate", "2", "--no-secondary",
              "--no-cpu-baseline", "--no-cold"]
    env = dict(os.environ, LC_BENCH_TEST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    for exchange in ("count", "mask"):
        two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                              "127.0.0.1", "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--exchange",
                              exchange] + common, capture_output=True, text=True, timeout=900, cwd=root, env=env)
        assert two.returncode == 0, two.stderr[-3000:]
        line = two.stdout.strip().splitlines()[-1]
        assert len(line) <= 4096, len(line)
        d2 = json.loads(line)
        assert d2["n_gpus"] == 2 and d2["scaling"] == "strong" and d2["steps"] == 2
        assert d2["config"]["hits"] == d1["config"]["hits"] > 0
        assert d2["config"]["rows_all_gpus"] == d1["config"]["rows_all_gpus"] == 3000000
        assert d2["config"]["scans_per_step"] == 4 and d2["ms_per_step"] > 0 and d2["value"] > 0
        assert "scaling_model" in d2 and d2["roofline"]["kernel"].startswith("k_like")
        # round 6: the exchange runs through the library's own communicator by default (here its shared-memory test backend: two
        # ranks on one GPU), proven by a known-answer all-reduce before it is used
        assert d2["config"]["exchange_by"].startswith("lc_comm"), d2["config"]["exchange_by"]


def _build_c(tmp_path, name):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    libdir = os.path.join(root, "liquid_cache_amd")
    subprocess.run(["gcc", "-std=gnu11", "-O1", "-Wall", "-Werror", "-pthread", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_abi", name + ".c"), "-o", exe, "-L", libdir, "-l:libliquid_cache_amd.so",
                    "-Wl,-rpath," + libdir], check=True)
    return exe


def test_rowgroup_reader_c_program(tmp_path, product_lib):
    """tests/c_abi/rowgroup_reader.c: the reader's loop at the reference's granularity through the plain C ABI — per row group
    the sparse pipeline (eval_hits -> filter_hits -> gathers from the list) and the mask form, four threads on own streams,
    every returned value checked against a plain C loop over the generated table."""
    import subprocess
    exe = _build_c(tmp_path, "rowgroup_reader")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rowgroup reader ok" in r.stdout


@pytest.mark.parametrize("like_path", [4, 0])
def test_index_builder_chunk_cuts_inside_escape_runs(product_lib, oracle, like_path):
    """k_flat_build walks a value in 16-byte chunks of its COMPRESSED bytes and enters the FSST stream at every cut: the parity
    of the run of 255s in front of a cut decides whether it falls between an escape marker and its literal, and the bigram
    across the cut needs the last byte of the symbol (or the literal) before it.  Values built so that runs of escaped bytes —
    0xFF as data included: marker 255 + literal 255 — lie across every cut position, against a symbol table trained on
    OTHER text (most bytes travel as escapes, two compressed bytes each).  A bigram the builder loses is a value the index
    never offers as a candidate: every needle below occurs in the data, so a lost bit is a lost hit."""
    lo = oracle
    rng = np.random.default_rng(20250926)
    other = [b"mail.google.com/inbox", b"yandex.ru/search?text=", b"http://example.org/index.php", b"aaaabbbbccccdddd"] * 50
    o, dt, _ = lo.strings_to_arrow(other)
    st = lo.fsst_train(o, dt)
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1, like_path=like_path or None).build()
    try:
        cache.set_symbol_table(7100, lo.symtab_bytes(st))
        ids, cases = [], []
        for b in range(6):
            pool = []
            for k in range(180):
                v = bytearray()
                # plain text the table compresses (1 byte per several), then escaped stretches of every length 0..11 so that the
                # cut positions 16, 32, 48, ... of the compressed stream land on every phase of marker / literal pairs
                v += other[int(rng.integers(len(other)))][: int(rng.integers(0, 22))]
                for _ in range(int(rng.integers(1, 9))):
                    run = int(rng.integers(0, 12))
                    kind = rng.random()
                    if kind < 0.4:
                        v += b"\xff" * run
                    elif kind < 0.8:
                        v += bytes(rng.integers(128, 256, size=run, dtype=np.uint8).tobytes())
                    else:
                        v += bytes([0xFF, int(rng.integers(1, 255))] * (run // 2))
                    v += [b"go", b"ogle", b"ma", b"il", b"", b"x"][int(rng.integers(6))]
                pool.append(bytes(v) + bytes([k & 0x7F, b]))  # (distinct values)
            rows = [pool[int(i)] for i in rng.integers(0, len(pool), size=700)]
            liquid, _ = lo.encode_byte_view(rows, st=st, fingerprints=True, arrow_type=lo.BT_BINARY)
            eid = lc.ParquetArrayID.new(31, b, 3, 0)
            cache.stage([eid], [liquid], [7100])
            ids.append(eid)
            cases.append((rows, liquid))
        scan = cache.scan(ids)
        offs = scan.segment_offsets
        needles = []
        for rows, _ in cases:                               # 2..6-byte windows out of the values, biased to the escaped parts
            for _ in range(14):
                v = rows[int(rng.integers(len(rows)))]
                ff = [i for i in range(len(v) - 1) if v[i] >= 128]
                a = ff[int(rng.integers(len(ff)))] if ff and rng.random() < 0.8 else int(rng.integers(max(len(v) - 1, 1)))
                a = max(0, a - int(rng.integers(0, 3)))
                nd = bytes(c for c in v[a: a + int(rng.integers(2, 7))] if c not in b"%_\\")
                if len(nd) >= 2:
                    needles.append(nd)
        needles += [b"\xff\xff", b"\xff\xff\xff", b"go\xff", b"\xffma", b"le\xff\xff"]
        n_hits = 0
        for nd in needles:
            expr = lc.LiquidExpr.try_new("like", b"%" + nd + b"%", pa.binary(), lc.CacheExpression.SUBSTRING_SEARCH)
            mask, counts = scan.eval_to_host(expr)
            bits = np.unpackbits(mask.view(np.uint8), bitorder="little")
            for b, (rows, liquid) in enumerate(cases):
                got = bits[int(offs[b]) * 64: int(offs[b]) * 64 + len(rows)].astype(bool)
                want = np.array([nd in r for r in rows])
                assert np.array_equal(got, want), (like_path, b, nd, int(got.sum()), int(want.sum()))
                assert int(counts[b]) == int(want.sum())
                n_hits += int(want.sum())
        assert n_hits > 500
        if like_path == 4:
            assert "k_like_flat" in scan.explain(lc.LiquidExpr.try_new("like", b"%\xff\xff%", pa.binary(),
                                                                       lc.CacheExpression.SUBSTRING_SEARCH))
    finally:
        cache.close()
