"""Round 6: the headline in the reference's call shape.

* LC_OPT_LIKE_INDEX_ASYNC: a scan's first LIKE is answered from the entry-level index while the scan-level index is built by
  the context's builder thread; results are identical BEFORE, DURING and AFTER the build (the reference keeps its prefilter
  construction out of the read path the same way: byte_view_array/conversions.rs:353-355, comparisons.rs:159-183).
* LC_OPT_SCAN_CACHE: lc_scan_create over an entry-id list seen before returns the kept scan — never a stale one: replacing or
  evicting an entry invalidates it (the reference's reader names entries per query, liquid_cache_reader.rs:264-339).
* lc_scan_eval_count_groups / lc_eval_predicate_row_groups: many row groups per call with per-row-group counts
  (liquid_stream.rs:358-430) == the per-entry results of the oracle summed per group.
"""
import os
import sys

import numpy as np
import pyarrow as pa
import pytest

import liquid_cache_amd as lc
from liquid_cache_amd import _native as N

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_data as fz  # noqa: E402
from test_gpu_round4 import RUNS, _want_full, grouped_cases  # noqa: E402,F401  (the grouped-entries fixture)

pytestmark = pytest.mark.gpu
HINT = lc.CacheExpression.SUBSTRING_SEARCH
NEEDLES = [b"google", b"index.php?id=1", b"zzzzqqq", b"mail", b"a", b"#21", b"yandex.google", b"//"]


def _stage(cache, lo, cases, file_id=40, path0=7600):
    ids, flat = [], []
    for r_i, (st, entries) in enumerate(cases):
        path = path0 + r_i
        cache.set_symbol_table(path, lo.symtab_bytes(st))
        for e_i, (rows, liquid) in enumerate(entries):
            eid = lc.ParquetArrayID.new(file_id, r_i, 5, e_i)
            cache.stage([eid], [liquid], [path])
            ids.append(eid)
            flat.append((rows, liquid, st))
    return ids, flat


def _oracle_bits(lo, flat, op, pattern):
    return [_want_full(lo, liquid, st, lo.OP_NAMES[op], pattern, None, len(rows)) for rows, liquid, st in flat]


def _scan_bits(scan, mask, lens):
    bits = np.unpackbits(mask.view(np.uint8), bitorder="little").astype(bool)
    offs = scan.segment_offsets
    return [bits[int(offs[b]) * 64: int(offs[b]) * 64 + n] for b, n in enumerate(lens)]


def _async_cache(options=None):
    b = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1).with_option(N.OPT_LIKE_INDEX_ASYNC, 1)
    for k, v in (options or {}).items():
        b = b.with_option(k, v)
    return b.build()


def test_results_identical_before_during_after_async_build(product_lib, oracle, grouped_cases):
    lo = oracle
    cache = _async_cache()
    try:
        ids, flat = _stage(cache, lo, grouped_cases)
        lens = [len(c[0]) for c in flat]
        want = {}
        for nd in NEEDLES:
            for op in ("like", "not_like"):
                want[(nd, op)] = _oracle_bits(lo, flat, op, b"%" + nd + b"%")
        for round_ in range(3):  # a fresh scan each round: the first evaluations race the builder
            scan = cache.scan(ids)
            phases = []
            for phase in ("before/during", "during", "after"):
                if phase == "after":
                    scan.index_wait()
                for nd in NEEDLES:
                    for op in ("like", "not_like"):
                        expr = lc.LiquidExpr.try_new(op, b"%" + nd + b"%", pa.binary(), HINT)
                        mask, counts = scan.eval_to_host(expr)
                        got = _scan_bits(scan, mask, lens)
                        for b, w in enumerate(want[(nd, op)]):
                            assert np.array_equal(got[b], w), (round_, phase, nd, op, b, int(got[b].sum()), int(w.sum()))
                            assert int(counts[b]) == int(w.sum())
                phases.append(phase)
            info = scan.info()
            assert int(info.index_bytes) > 0, "the scan-level index never arrived"
            how = scan.explain(lc.LiquidExpr.try_new("like", b"%zzzzqqq%", pa.binary(), HINT))
            assert how.startswith("k_like_flat"), how
            # 1-byte needle: the unigram index arrives the same way
            one = lc.LiquidExpr.try_new("like", b"%a%", pa.binary(), HINT)
            m0, _ = scan.eval_to_host(one)
            scan.index_wait()
            m1, _ = scan.eval_to_host(one)
            assert np.array_equal(m0, m1)
            assert int(scan.info().unigram_index_bytes) > 0
            assert scan.explain(one).startswith("k_like_scanall<unigram>")
            if round_ == 1:
                # the last round builds everything again on a scan that is really new
                cache.set_option(N.OPT_SCAN_CACHE, 0)
                cache.set_option(N.OPT_LIKE_INDEX_CACHE, 0)
            scan.close()
    finally:
        cache.close()


def test_scan_destroyed_while_its_index_is_being_built(product_lib, oracle, grouped_cases):
    """lc_scan_destroy waits for the builder (which reads the scan); nothing leaks, the next scan adopts the finished index."""
    lo = oracle
    cache = _async_cache()
    try:
        ids, flat = _stage(cache, lo, grouped_cases, file_id=41, path0=7700)
        lens = [len(c[0]) for c in flat]
        expr = lc.LiquidExpr.try_new("like", b"%index.php?id=1%", pa.binary(), HINT)
        want = _oracle_bits(lo, flat, "like", b"%index.php?id=1%")
        for k in range(6):
            scan = cache.scan(ids)
            mask, _ = scan.eval_to_host(expr)
            got = _scan_bits(scan, mask, lens)
            for b, w in enumerate(want):
                assert np.array_equal(got[b], w), (k, b)
            scan.close()  # at once: the build of round 0 is most likely still in flight
        scan = cache.scan(ids)
        scan.eval_to_host(expr)
        scan.index_wait()
        assert int(scan.info().index_bytes) > 0
        assert scan.info().index_build_pending == 0
        scan.close()
    finally:
        cache.close()


def test_scan_cache_returns_the_kept_scan_and_never_a_stale_one(product_lib, oracle, grouped_cases):
    lo = oracle
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1).build()
    try:
        ids, flat = _stage(cache, lo, grouped_cases, file_id=42, path0=7800)
        lens = [len(c[0]) for c in flat]
        expr = lc.LiquidExpr.try_new("like", b"%google%", pa.binary(), HINT)
        want = _oracle_bits(lo, flat, "like", b"%google%")

        def check(scan, want_bits):
            mask, counts = scan.eval_to_host(expr)
            got = _scan_bits(scan, mask, lens)
            for b, w in enumerate(want_bits):
                assert np.array_equal(got[b], w), b
                assert int(counts[b]) == int(w.sum())
        s1 = cache.scan(ids)
        h1 = s1._h.value
        check(s1, want)
        s1.close()
        s2 = cache.scan(ids)
        assert s2._h.value == h1, "the kept scan was not handed out again"
        check(s2, want)
        # two scans over one list at a time: the second is a new one, both are right
        s3 = cache.scan(ids)
        assert s3._h.value != h1
        check(s3, want)
        s3.close()
        s2.close()
        # a different list (a prefix) is a different scan
        sp = cache.scan(ids[:5])
        assert sp.entries == 5
        sp.close()
        # replace ONE entry (same id, other rows): the kept scans that hold it are gone, the next scan sees the new data
        k = 2
        rows_new = [r if i % 2 else (None if r is None else r + b"google") for i, r in enumerate(flat[k][0])]
        liquid_new, _ = lo.encode_byte_view(rows_new, st=flat[k][2], fingerprints=True, arrow_type=lo.BT_BINARY)
        cache.stage([ids[k]], [liquid_new], [7800])
        flat2 = list(flat)
        flat2[k] = (rows_new, liquid_new, flat[k][2])
        want2 = _oracle_bits(lo, flat2, "like", b"%google%")
        assert int(want2[k].sum()) != int(want[k].sum())
        s4 = cache.scan(ids)
        check(s4, want2)
        # ... and a scan that was in a caller's hands while its entry was replaced is not kept when it is given back
        liquid_back = flat[k][1]
        cache.stage([ids[k]], [liquid_back], [7800])
        check(s4, want2)  # (a live scan keeps the blob it captured, like the reference's Arc clone)
        h4 = s4._h.value
        s4.close()
        s5 = cache.scan(ids)
        check(s5, want)
        s5.close()
        del h4
        # evicting an entry: the list is no longer complete
        cache.evict([ids[0]])
        with pytest.raises(N.LiquidCacheError) as ei:
            cache.scan(ids)
        assert ei.value.status == N.LC_NOT_STAGED
        s6 = cache.scan(ids[1:])
        mask, _ = s6.eval_to_host(expr)
        got = _scan_bits(s6, mask, lens[1:])
        for b, w in enumerate(want[1:]):
            assert np.array_equal(got[b], w)
        s6.close()
        # LC_OPT_SCAN_CACHE = 0: every destroy frees
        cache.set_option(N.OPT_SCAN_CACHE, 0)
        s7 = cache.scan(ids[1:])
        s7.close()
    finally:
        cache.close()


def test_row_groups_per_call_against_oracle(product_lib, oracle, grouped_cases):
    lo = oracle
    cache = _async_cache()
    try:
        ids, flat = _stage(cache, lo, grouped_cases, file_id=43, path0=7900)
        lens = [len(c[0]) for c in flat]
        n = len(ids)
        rng = np.random.default_rng(11)
        groupings = [[n], list(range(1, n + 1)), [5, 17, n], [0, 3, 3, n]]  # one group; one per entry; row groups; empty groups
        for nd in (b"google", b"zzzzqqq", b"a", b"index.php?id=1"):
            for op in ("like", "not_like"):
                pattern = b"%" + nd + b"%"
                expr = lc.LiquidExpr.try_new(op, pattern, pa.binary(), HINT)
                want = _oracle_bits(lo, flat, op, pattern)
                per_entry = np.array([int(w.sum()) for w in want], np.uint64)
                for ends in groupings:
                    counts, total, mask = cache.eval_predicate_row_groups(ids, ends, expr, want_mask=True)
                    b0 = 0
                    for g, e in enumerate(ends):
                        assert int(counts[g]) == int(per_entry[b0:e].sum()), (nd, op, ends, g)
                        b0 = e
                    assert total == int(per_entry.sum())
                    bits = np.unpackbits(mask.view(np.uint8), bitorder="little").astype(bool)
                    off = 0
                    for b, w in enumerate(want):
                        assert np.array_equal(bits[off * 64: off * 64 + lens[b]], w), (nd, op, b)
                        off += (lens[b] + 63) // 64
        # integers through the same call: row groups of a fixed-width column, one and two fused predicates
        vals = rng.integers(-1000, 1000, size=9 * 8192 + 77, dtype=np.int64)
        iids = []
        for b in range(10):
            eid = lc.ParquetArrayID.new(44, b // 4, 1, b % 4)
            cache.insert(eid, pa.array(vals[b * 8192:(b + 1) * 8192]))
            iids.append(eid)
        ends = [4, 8, 10]
        gt = lc.LiquidExpr.try_new(">", 12, pa.int64())
        lt = lc.LiquidExpr.try_new("<", 500, pa.int64())
        for exprs, fn in (([gt], lambda v: v > 12), ([gt, lt], lambda v: (v > 12) & (v < 500))):
            counts, total, _ = cache.eval_predicate_row_groups(np.asarray([int(e) for e in iids], np.uint64), ends, exprs)
            b0 = 0
            for g, e in enumerate(ends):
                assert int(counts[g]) == int(fn(vals[b0 * 8192:e * 8192]).sum())
                b0 = e
            assert total == int(fn(vals).sum())
        # the device-resident form on a scan, with per-entry counts asked for as well
        import ctypes as C
        scan = cache.scan(iids)
        d_g, d_c = C.c_void_p(), C.c_void_p()
        N.check(cache._lib.lc_device_alloc(cache.handle, 8 * len(ends), C.byref(d_g)), cache.handle)
        N.check(cache._lib.lc_device_alloc(cache.handle, 4 * len(iids), C.byref(d_c)), cache.handle)
        scan.eval_count_groups(gt, ends, d_g.value, counts_ptr=d_c.value)
        hg, hc = np.zeros(len(ends), np.uint64), np.zeros(len(iids), np.uint32)
        N.check(cache._lib.lc_device_to_host(cache.handle, hg.ctypes.data_as(C.c_void_p), d_g, hg.nbytes, None), cache.handle)
        N.check(cache._lib.lc_device_to_host(cache.handle, hc.ctypes.data_as(C.c_void_p), d_c, hc.nbytes, None), cache.handle)
        assert hc.tolist() == [int((vals[b * 8192:(b + 1) * 8192] > 12).sum()) for b in range(10)]
        assert hg.tolist() == [int(hc[:4].sum()), int(hc[4:8].sum()), int(hc[8:].sum())]
        # bad groupings are refused
        for bad in ([3, 2, 10], [4, 8], [4, 11]):
            with pytest.raises(N.LiquidCacheError):
                scan.eval_count_groups(gt, bad, d_g.value)
        cache._lib.lc_device_free(cache.handle, d_g)
        cache._lib.lc_device_free(cache.handle, d_c)
        scan.close()
        # an absent entry: the reference's `None`
        with pytest.raises(N.LiquidCacheError) as ei:
            cache.eval_predicate_row_groups([int(iids[0]), 123456789], [2], gt)
        assert ei.value.status == N.LC_NOT_STAGED
    finally:
        cache.close()


def test_async_build_does_not_evict_until_the_scan_is_hot(product_lib, oracle, grouped_cases):
    """Under a budget of ONE index: column A has the index; column B's first evaluations run on the entry-level index and
    leave A's cached index alone; after kEvictAfterEvals (8) evaluations B takes the room.  Results equal the oracle's
    throughout."""
    lo = oracle
    probe = _async_cache()
    try:
        ids, flat = _stage(probe, lo, grouped_cases, file_id=45, path0=8000)
        s = probe.scan(ids)
        expr = lc.LiquidExpr.try_new("like", b"%zzzzqqq%", pa.binary(), HINT)
        s.eval_to_host(expr)
        s.index_wait()
        index_bytes = int(s.info().index_bytes)
        assert index_bytes > 0
        s.close()
    finally:
        probe.close()
    cache = _async_cache({N.OPT_LIKE_INDEX_BUDGET_BYTES: index_bytes * 3 // 2})
    try:
        ids_a, flat = _stage(cache, lo, grouped_cases, file_id=46, path0=8100)
        ids_b, _ = _stage(cache, lo, grouped_cases, file_id=47, path0=8200)
        lens = [len(c[0]) for c in flat]
        want = _oracle_bits(lo, flat, "like", b"%zzzzqqq%")

        def query(ids):
            sc = cache.scan(ids)
            mask, _ = sc.eval_to_host(expr)
            got = _scan_bits(sc, mask, lens)
            for b, w in enumerate(want):
                assert np.array_equal(got[b], w)
            sc.index_wait()
            ib = int(sc.info().index_bytes)
            sc.close()
            return ib
        assert query(ids_a) == index_bytes
        seen = [query(ids_b) for _ in range(7)]
        assert seen == [0] * 7, seen  # A's cached index is not evicted for a scan that has not proven hot
        assert query(ids_a) == index_bytes
        later = [query(ids_b) for _ in range(4)]
        assert later[-1] == index_bytes, later  # ... and is once B has served 8 evaluations
    finally:
        cache.close()


def _read_partitioned(scan, d_hits, d_n, cap):
    """(records of partition 0, 1, ... concatenated — the order the consuming calls read them in —, per-partition counts)."""
    ctrs = scan._from_dev(d_n, np.uint64, N.HITS_PARTITIONS * N.HITS_COUNTER_STRIDE)[::N.HITS_COUNTER_STRIDE]
    stride = cap // N.HITS_PARTITIONS
    buf = scan._from_dev(d_hits, np.uint64, cap)
    parts = [buf[p * stride: p * stride + min(int(ctrs[p]), stride)] for p in range(N.HITS_PARTITIONS)]
    return np.concatenate(parts) if parts else np.zeros(0, np.uint64), ctrs


def test_partitioned_hit_lists_against_oracle(product_lib, oracle, grouped_cases):
    """LC_HITS_PARTITIONED: the list in 16 partitions (producers claim space on 16 addresses instead of one).  The records of
    all partitions are exactly the set bits of the oracle's mask — from k_like_flat itself and from the mask path —, the filter
    on a partitioned list keeps exactly the oracle's survivors, the gathers read the partitions as one list in partition order
    (row i of the output = record i of that order), lc_hits_compact gives that order back as a contiguous list."""
    lo = oracle
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1).build()
    try:
        ids, flat = _stage(cache, lo, grouped_cases, file_id=48, path0=8300)
        lens = [len(c[0]) for c in flat]
        scan = cache.scan(ids)
        offs = scan.segment_offsets
        n_bits = int(scan.mask_words) * 64
        # (a partition holds capacity / 16 records and a small launch fills only as many partitions as it has workgroups: sized
        # so that ONE partition can take every row — a caller sizes for its selectivity and retries on overflow)
        cap = int(scan.rows) * N.HITS_PARTITIONS
        lib, ctx = cache._lib, cache.handle
        ptrs = []

        def dev(nbytes):
            p = scan._dev(nbytes)
            ptrs.append(p)
            return p
        d_hits, d_hits2, d_flat = dev(cap * 8), dev(cap * 8), dev(cap * 8)
        d_n, d_n2 = dev(2048), dev(2048)
        d_nflat = dev(8)
        d_first = dev(max(scan.entries, 1) * 4)
        d_mask = dev(max(int(scan.mask_words), 1) * 8)
        d_total = dev(8)
        checked = 0
        for nd, op in ((b"zzzzqqq", "like"), (b"index.php?id=1", "like"), (b"google", "like"), (b"a", "like"), (b"#21", "not_like"),
                       (b"yandex.google", "like"), (b"//", "like")):
            pattern = b"%" + nd + b"%"
            expr = lc.LiquidExpr.try_new(op, pattern, pa.binary(), HINT)
            want = _oracle_bits(lo, flat, op, pattern)
            want_set = set()
            for b, w in enumerate(want):
                want_set.update((b << 32) | int(r) for r in np.flatnonzero(w))
            for from_mask in (False, True):
                N.check(lib.lc_host_to_device(ctx, d_first, np.full(max(scan.entries, 1), 0xFFFFFFFF, np.uint32).ctypes.data_as(
                    __import__("ctypes").c_void_p), max(scan.entries, 1) * 4, None), ctx)
                if from_mask:
                    scan.eval_count(expr, d_mask.value, d_total.value)
                    scan.mask_to_hits(d_mask.value, d_hits.value, cap, d_n.value, d_first.value, partitioned=True)
                else:
                    scan.eval_hits(expr, d_hits.value, cap, d_n.value, hit_first_ptr=d_first.value, total_out_ptr=d_total.value,
                                   partitioned=True)
                N.check(lib.lc_stream_synchronize(ctx, None), ctx)
                recs, ctrs = _read_partitioned(scan, d_hits, d_n, cap)
                total = int(scan._from_dev(d_total, np.uint64, 1)[0])
                assert int(ctrs.sum()) == len(recs) == total == len(want_set), (nd, op, from_mask, int(ctrs.sum()), len(want_set))
                assert set(int(x) for x in recs) == want_set, (nd, op, from_mask)
                # an entry's records are contiguous and ascending inside ONE partition; hit_first is its position in the buffer
                first = scan._from_dev(d_first, np.uint32, scan.entries)
                buf = scan._from_dev(d_hits, np.uint64, cap)
                for b, w in enumerate(want):
                    rows = np.flatnonzero(w)
                    if len(rows) == 0:
                        continue
                    f = int(first[b])
                    got = buf[f: f + len(rows)]
                    assert [int(x) for x in got] == [(b << 32) | int(r) for r in rows], (nd, op, from_mask, b)
                # the compaction: the same records, partition order, contiguous
                scan.hits_compact(d_hits.value, d_n.value, cap, d_flat.value, cap, d_nflat.value)
                N.check(lib.lc_stream_synchronize(ctx, None), ctx)
                nflat = int(scan._from_dev(d_nflat, np.uint64, 1)[0])
                assert nflat == len(recs)
                assert scan._from_dev(d_flat, np.uint64, nflat).tolist() == recs.tolist()
                checked += 1
            # the next conjunct on the partitioned list (same column, another pattern) == the oracle on the listed rows
            e2 = lc.LiquidExpr.try_new("like", b"%ru%", pa.binary(), HINT)
            w2 = _oracle_bits(lo, flat, "like", b"%ru%")
            scan.filter_hits(e2, d_hits.value, d_n.value, cap, d_hits2.value, cap, d_n2.value, partitioned=True)
            N.check(lib.lc_stream_synchronize(ctx, None), ctx)
            recs2, _ = _read_partitioned(scan, d_hits2, d_n2, cap)
            surv = set(x for x in want_set if w2[x >> 32][x & 0xFFFFFFFF])
            assert set(int(x) for x in recs2) == surv and len(recs2) == len(surv), (nd, op)
            # the byte-view gather reads the partitions as one list: row i = record i of the partition order
            if 0 < len(recs) <= 4000:
                k = len(recs)
                # (capacity_rows is the LIST's capacity — it fixes the partitions' places — and bounds the outputs as well)
                cap_b = cap * 128 + (1 << 20)
                d_views, d_valid, d_data, d_nb = scan._dev(cap * 16), scan._dev(cap), scan._dev(cap_b), scan._dev(8)
                try:
                    scan.gather_bytes_hits(d_hits.value, d_n.value, cap, d_views.value, d_data.value, cap_b, d_nb.value,
                                           d_valid.value, slotted=True, partitioned=True)
                    N.check(lib.lc_stream_synchronize(ctx, None), ctx)
                    views = scan._from_dev(d_views, np.uint8, k * 16).reshape(-1, 16)
                    data = scan._from_dev(d_data, np.uint8, cap_b).tobytes()
                    for i in range(0, k, max(1, k // 200)):
                        b, r = int(recs[i]) >> 32, int(recs[i]) & 0xFFFFFFFF
                        v = flat[b][0][r]
                        ln = int(views[i, :4].view(np.int32)[0])
                        assert ln == len(v), (i, ln, len(v))
                        got = views[i, 4:4 + ln].tobytes() if ln <= 12 else data[int(views[i, 12:16].view(np.int32)[0]):][:ln]
                        assert got == v, (nd, i)
                finally:
                    for p in (d_views, d_valid, d_data, d_nb):
                        lib.lc_device_free(ctx, p)
        assert checked == 14
        for p in ptrs:
            lib.lc_device_free(ctx, p)
        scan.close()
        # fixed width: hits of a predicate through the mask path, partitioned; the gather projects the same column
        rng = np.random.default_rng(3)
        vals = rng.integers(-10**6, 10**6, size=5 * 8192 - 77, dtype=np.int64)
        iids = []
        for b in range(5):
            eid = lc.ParquetArrayID.new(49, 0, 1, b)
            cache.insert(eid, pa.array(vals[b * 8192:(b + 1) * 8192]))
            iids.append(eid)
        s2 = cache.scan(iids)
        gt = lc.LiquidExpr.try_new(">", 900000, pa.int64())
        cap2 = int(s2.rows) * N.HITS_PARTITIONS
        d_h, d_c = s2._dev(cap2 * 8), s2._dev(2048)
        s2.eval_hits(gt, d_h.value, cap2, d_c.value, partitioned=True)
        N.check(lib.lc_stream_synchronize(ctx, None), ctx)
        recs, ctrs = _read_partitioned(s2, d_h, d_c, cap2)
        keep = np.flatnonzero(vals > 900000)
        assert sorted((int(x) >> 32) * 8192 + (int(x) & 0xFFFFFFFF) for x in recs) == keep.tolist()
        d_v = s2._dev(len(recs) * 8 + 64)
        s2.gather_fixed_hits(d_h.value, d_c.value, cap2, d_v.value, partitioned=True)
        N.check(lib.lc_stream_synchronize(ctx, None), ctx)
        got = s2._from_dev(d_v, np.int64, len(recs))
        assert got.tolist() == [int(vals[(int(x) >> 32) * 8192 + (int(x) & 0xFFFFFFFF)]) for x in recs]
        for p in (d_h, d_c, d_v):
            lib.lc_device_free(ctx, p)
        s2.close()
    finally:
        cache.close()


# ------------------------------------------------------------------------------------------------------------------
# lc_scan_eval_filter chains: a conjunction over several fixed-width columns in one launch
# ------------------------------------------------------------------------------------------------------------------
def _chain_column(rng, kind, n, W, null_frac, all_null):
    """(arrow array, numpy values, validity) of one entry: values span exactly W bits above a random base."""
    span = (1 << W) - 1
    if kind == "date32":
        base = int(rng.integers(0, 20000))
        vals = (base + rng.integers(0, span + 1, size=n)).astype(np.int32)
        vals[0], vals[-1] = base, base + span
        dtype = pa.date32()
    elif kind == "int32":
        base = int(rng.integers(-100000, 100000))
        vals = (base + rng.integers(0, span + 1, size=n)).astype(np.int32)
        vals[0], vals[-1] = base, base + span
        dtype = pa.int32()
    else:
        base = int(rng.integers(-(1 << 40), 1 << 40))
        vals = (base + rng.integers(0, span + 1, size=n)).astype(np.int64)
        vals[0], vals[-1] = base, base + span
        dtype = pa.int64()
    valid = np.ones(n, bool)
    if all_null:
        valid[:] = False
    elif null_frac > 0:
        valid = rng.random(n) >= null_frac
        valid[0] = valid[-1] = True  # (the extremes keep the entry's width at W)
    arr = pa.array(vals, type=dtype, mask=~valid) if not valid.all() else pa.array(vals, type=dtype)
    return arr, vals, valid, base, dtype


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_chain_against_oracle(gpu_cache, oracle, seed):
    """lc_scan_eval_filter over 2-4 columns on u32 / u64 lanes (k_fixed_chain, or k_fixed_chain_lds in a -DLC_X_CHAIN_LDS=1 build): every width 1..16, entries of ragged
    lengths (1 row, one block, an odd number of blocks, tails), nulls, an all-null entry, literals outside an entry's range
    (constant outcomes), fused range pairs, Ne, with and without a selection in front — final mask, per-entry counts and the
    fused COUNT(*) against the oracle's evaluation of the same Liquid bytes, conjunct by conjunct."""
    import ctypes as C
    from liquid_cache_amd.pushdown import CompiledFilter
    lo = oracle
    rng = np.random.default_rng(1000 + seed)
    lens = [8192, 1, 1024, 1025, 3000, 2048 + 70, 5000, 8192, 64, 4097]
    kinds = [["date32", "int64", "int64"], ["int32", "int64"], ["int64", "date32", "int32", "int64"]][seed - 1]
    n_cols = len(kinds)
    widths = [[int(rng.integers(1, 17)) for _ in lens] for _ in range(n_cols)]
    for c in range(n_cols):  # every width somewhere
        for k in range(len(lens)):
            widths[c][k] = 1 + (k * 5 + c * 7 + seed * 3) % 16
    ids, liquids, info = [], [], []
    for c, kind in enumerate(kinds):
        ids.append([])
        liquids.append([])
        info.append([])
        for k, n in enumerate(lens):
            all_null = (c == 1 and k == 4)
            arr, vals, valid, base, dtype = _chain_column(rng, kind, n, widths[c][k], 0.2 if (k + c) % 3 == 0 else 0.0, all_null)
            eid = lc.ParquetArrayID.new(60 + seed, 0, c, k)
            gpu_cache.insert(eid, arr)
            ids[c].append(eid)
            liquids[c].append(gpu_cache.transcode(arr))
            info[c].append((base, widths[c][k], dtype))
    scans = [gpu_cache.scan(ids[c]) for c in range(n_cols)]
    E = lc.LiquidExpr.try_new
    import datetime

    def lit(c, v):
        dtype = info[c][0][2]
        return datetime.date(1970, 1, 1) + datetime.timedelta(days=int(v)) if dtype == pa.date32() else int(v)

    for variant in range(4):
        # per column: one predicate or a fused pair; literals around the middle of SOME entry's range, so that other entries
        # see them outside theirs (constant outcomes)
        steps, oracle_steps = [], []
        for c in range(n_cols):
            base, W, dtype = info[c][(variant * 3 + c) % len(lens)]
            mid = base + (1 << W) // 2
            if (variant + c) % 3 == 0:
                ops = [("ge", mid - (1 << W) // 4), ("lt", mid + (1 << W) // 4 + 1)]
            elif (variant + c) % 3 == 1:
                ops = [(["lt", "le", "gt", "ge"][(variant + c) % 4], mid)]
            else:
                ops = [(["eq", "ne"][(variant + c) % 2], mid)]
            exprs = [E(o, lit(c, v), dtype) for o, v in ops]
            assert all(e is not None for e in exprs)
            steps.append((scans[c], exprs))
            oracle_steps.append((c, ops))
        cf = CompiledFilter.from_conjunction(steps)
        for with_sel in (False, True):
            words = int(scans[0].mask_words)
            sel = None
            if with_sel:
                sel_bits = [rng.random(n) < (0.0 if k == 2 else 0.3) for k, n in enumerate(lens)]
                sel_bits[5][:2048] = False  # a whole pass without a selected row
                sel = np.zeros(words, np.uint64)
                for k, n in enumerate(lens):
                    w0 = int(scans[0].segment_offsets[k])
                    packed = np.packbits(sel_bits[k], bitorder="little")
                    sel[w0: w0 + (n + 63) // 64] = np.frombuffer(packed.tobytes() + b"\0" * (-len(packed) % 8), np.uint64)
            lib, ctx = scans[0]._lib, gpu_cache.handle
            bufs = [C.c_void_p() for _ in range(5)]
            for b, nbytes in zip(bufs, [words * 8, words * 8, len(lens) * 4, 8, words * 8]):
                N.check(lib.lc_device_alloc(ctx, max(nbytes, 8), C.byref(b)), ctx)
            if sel is not None:
                N.check(lib.lc_host_to_device(ctx, bufs[4], sel.ctypes.data_as(C.c_void_p), sel.size * 8, None), ctx)
            final = cf.run(bufs[0].value, bufs[1].value, bufs[2].value, bufs[4].value if sel is not None else 0, bufs[3].value)
            mask = np.zeros(words, np.uint64)
            counts = np.zeros(len(lens), np.uint32)
            total = np.zeros(1, np.uint64)
            N.check(lib.lc_device_to_host(ctx, mask.ctypes.data_as(C.c_void_p), C.c_void_p(final), words * 8, None), ctx)
            N.check(lib.lc_device_to_host(ctx, counts.ctypes.data_as(C.c_void_p), bufs[2], counts.size * 4, None), ctx)
            N.check(lib.lc_device_to_host(ctx, total.ctypes.data_as(C.c_void_p), bufs[3], 8, None), ctx)
            for b in bufs:
                lib.lc_device_free(ctx, b)
            want_total = 0
            for k, n in enumerate(lens):
                w = None if sel is None else sel_bits[k]
                for c, ops in oracle_steps:
                    for o, v in ops:
                        r = lo.eval_predicate(liquids[c][k], lo.OP_NAMES[o], int(v), None)
                        hit = r.values if r.validity is None else (r.values & r.validity)
                        w = hit if w is None else (w & hit)
                w0 = int(scans[0].segment_offsets[k])
                got = np.unpackbits(mask[w0: w0 + (n + 63) // 64].view(np.uint8), bitorder="little")[:n].astype(bool)
                assert np.array_equal(got, w), (seed, variant, with_sel, k, widths)
                assert int(counts[k]) == int(w.sum()), (seed, variant, with_sel, k)
                want_total += int(w.sum())
            assert int(total[0]) == want_total
    for s in scans:
        s.close()
