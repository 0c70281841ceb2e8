"""Full-size (BASELINE.json configs[1]: 99,997,497 rows) properties of the HIP path, through the C ABI.

The oracle cannot cover 100 M rows in seconds, so these tests use what the domain offers at any size: per-entry counts
against an independent numpy / Python evaluation of regenerated sample batches, popcount(mask) == sum(counts),
complement laws, idempotence under selection chaining, and monotonicity of needles.
"""
import argparse
import os

import numpy as np
import pyarrow as pa
import pytest

import liquid_cache_amd as lc
from liquid_cache_amd import _native as N

pytestmark = pytest.mark.gpu

ROWS = 99_997_497
BS = 8192


def _args(**kw):
    """bench.py's own defaults (so that a new staging option can never be missing here), at the full ClickBench size."""
    import bench
    a = bench.parse_args([])
    a.rows, a.batch_size = ROWS, BS
    a.__dict__.update(kw)
    return a


def _popcount(words: np.ndarray) -> int:
    return int(np.unpackbits(words.view(np.uint8)).sum())


def _entry_bits(mask, scan, b, rows):
    off = int(scan.segment_offsets[b])
    return np.unpackbits(mask[off: off + (rows + 63) // 64].view(np.uint8), bitorder="little")[:rows].astype(bool)


@pytest.fixture(scope="module")
def bench_mod():
    import bench
    return bench


def test_int64_scan_full_size(product_lib, bench_mod):
    args = _args()
    n_batches = (ROWS + BS - 1) // BS
    threads = max(1, min(32, os.cpu_count() or 8))
    cache = lc.LiquidCacheBuilder.new().build()
    try:
        ids = bench_mod.stage_int_column(cache, lc, N, args, 0, n_batches, threads)
        scan = cache.scan(ids)
        assert scan.rows == ROWS and scan.entries == n_batches
        base = 4_000_000_000_000_000_000 >> (64 - args.int_bits)
        lit = base + (1 << (args.int_bits - 1))
        m_gt, c_gt = scan.eval_to_host(lc.LiquidExpr.try_new(">", lit, pa.int64()))
        m_le, c_le = scan.eval_to_host(lc.LiquidExpr.try_new("<=", lit, pa.int64()))
        assert _popcount(m_gt) == int(c_gt.sum()) and _popcount(m_le) == int(c_le.sum())
        assert int(c_gt.sum()) + int(c_le.sum()) == ROWS                       # complement (no nulls)
        assert not (m_gt & m_le).any()
        # regenerated sample batches, evaluated by numpy
        L = N.load()
        rng = np.random.default_rng(0)
        buf = np.zeros(BS, np.int64)
        for b in [0, n_batches - 1] + [int(x) for x in rng.integers(1, n_batches - 1, size=40)]:
            rows = min(BS, ROWS - b * BS)
            L.lc_synth_int64_batch(args.seed, b, rows, args.int_bits, base, buf.ctypes.data)
            want = buf[:rows] > lit
            assert int(c_gt[b]) == int(want.sum()), b
            assert _entry_bits(m_gt, scan, b, rows).tolist() == want.tolist(), b
        # chaining: the mask as selection of the same predicate is idempotent; of the complement it is empty
        m2, c2 = scan.eval_to_host(lc.LiquidExpr.try_new(">", lit, pa.int64()), selection=m_gt)
        assert (m2 == m_gt).all() and (c2 == c_gt).all()
        m3, c3 = scan.eval_to_host(lc.LiquidExpr.try_new("<=", lit, pa.int64()), selection=m_gt)
        assert int(c3.sum()) == 0 and not m3.any()
        # literals outside every batch's range: constant results without touching the data
        _, c_all = scan.eval_to_host(lc.LiquidExpr.try_new(">", -1, pa.int64()))
        assert int(c_all.sum()) == ROWS
        scan.close()
    finally:
        cache.close()


def test_url_like_scan_full_size(product_lib, bench_mod):
    args = _args()
    n_batches = (ROWS + BS - 1) // BS
    threads = max(1, min(32, os.cpu_count() or 8))
    cache = lc.LiquidCacheBuilder.new().build()
    try:
        ids = bench_mod.stage_url_column(cache, lc, N, args, 0, n_batches, threads)
        scan = cache.scan(ids)
        assert scan.rows == ROWS
        hint = lc.CacheExpression.SUBSTRING_SEARCH
        like = lambda pat, op="like": lc.LiquidExpr.try_new(op, pat.encode(), pa.string(), hint)
        m, c = scan.eval_to_host(like("%google%"))
        assert _popcount(m) == int(c.sum())
        # NOT LIKE is the complement here: every batch has fingerprint candidates for "google" (~44 % of its
        # dictionary), so the reference's zero-candidate quirk does not fire
        mn, cn = scan.eval_to_host(like("%google%", "not_like"))
        assert int(c.sum()) + int(cn.sum()) == ROWS and not (m & mn).any()
        # a longer needle selects a subset, a shorter one a superset
        m_long, c_long = scan.eval_to_host(like("%google.%"))
        m_short, c_short = scan.eval_to_host(like("%goog%"))
        assert not (m_long & ~m).any() and not (m & ~m_short).any()
        assert int(c_long.sum()) <= int(c.sum()) <= int(c_short.sum())
        # regenerated sample batches, evaluated by Python's substring test
        L = N.load()
        rng = np.random.default_rng(1)
        offs = np.zeros(BS + 1, np.int32)
        data = np.zeros(BS * 512, np.uint8)
        sample = [0, n_batches - 1] + [int(x) for x in rng.integers(1, n_batches - 1, size=30)]
        sample += [int(b) for b in np.nonzero(c)[0][:10]]           # and batches that do have matches
        for b in sample:
            rows = min(BS, ROWS - b * BS)
            n = L.lc_synth_url_batch(args.seed, b, rows, min(args.uniques, rows), args.needle_ppm, offs.ctypes.data, data.ctypes.data,
                                     data.size)
            raw = data[:n].tobytes()
            want = np.array([b"google" in raw[offs[i]:offs[i + 1]] for i in range(rows)])
            assert int(c[b]) == int(want.sum()), b
            assert _entry_bits(m, scan, b, rows).tolist() == want.tolist(), b
        # idempotence under chaining, and conjunction with an Eq predicate on the same column
        m2, c2 = scan.eval_to_host(like("%google%"), selection=m)
        assert (m2 == m).all() and (c2 == c).all()
        m3, c3 = scan.eval_to_host(like("%google%", "not_like"), selection=m)
        assert int(c3.sum()) == 0
        scan.close()
    finally:
        cache.close()
