"""Times of the sparse-result kernels over the 100 M-row URL column for a product-library build (LC_LIB_PATH), no result check
beyond the hit count: lc_scan_eval_hits (k_like_flat writing the hit list), lc_scan_filter_hits (`URL <> ''` over that list:
k_pred_hits), lc_scan_gather_bytes_hits.  HIP events around a loop of back-to-back calls on one stream (the calls are
asynchronous; a call's host time is below the kernels' durations).
usage: python scripts/time_sparse.py [--rows N] [--iters 30]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=99_997_497)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    args = bench.parse_args(["--rows", str(a.rows)])
    import torch
    import pyarrow as pa
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N
    cache = lc.LiquidCacheBuilder.new().build()
    n_batches = (args.rows + args.batch_size - 1) // args.batch_size
    ids = bench.stage_url_column(cache, lc, N, args, 0, n_batches, 16)
    scan = cache.scan(ids)
    like = lc.LiquidExpr.try_new("like", b"%google%", pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
    ne = lc.LiquidExpr.try_new("!=", "", pa.string())
    stream = torch.cuda.current_stream().cuda_stream
    cap = 1 << 21
    hits = torch.zeros(cap, dtype=torch.int64, device="cuda")
    hits2 = torch.zeros(cap, dtype=torch.int64, device="cuda")
    ctr = torch.zeros(4, dtype=torch.int64, device="cuda")  # n_hits, n_hits2, n_bytes
    total = torch.zeros(1, dtype=torch.int64, device="cuda")
    views = torch.zeros((cap, 2), dtype=torch.int64, device="cuda")
    data = torch.zeros(cap * 128 + (64 << 20), dtype=torch.uint8, device="cuda")
    p = ctr.data_ptr()

    def ev():
        scan.eval_hits(like, hits.data_ptr(), cap, p, 0, 0, 0, total.data_ptr(), stream, counters_zeroed=True)

    def count_only():
        scan.eval_count([like], 0, total.data_ptr(), 0, 0, stream)

    def flt():
        scan.filter_hits(ne, hits.data_ptr(), p, cap, hits2.data_ptr(), cap, p + 8, stream, counters_zeroed=True)

    def gat():
        scan.gather_bytes_hits(hits.data_ptr(), p, cap, views.data_ptr(), data.data_ptr(), min(data.numel(), (1 << 31) - 1), p + 16,
                               0, stream, counters_zeroed=True)

    def gat_slotted():
        scan.gather_bytes_hits(hits.data_ptr(), p, cap, views.data_ptr(), data.data_ptr(), min(data.numel(), (1 << 31) - 1), p + 16,
                               0, stream, counters_zeroed=True, slotted=True)

    def timed(fn, zero):
        ctr.zero_()
        ev()
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            if zero is not None:
                N.check(cache._lib.lc_device_memset(cache.handle, p + zero, 0, 8, stream), cache.handle)
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3

    tag = os.path.basename(os.environ.get("LC_LIB_PATH", "default")).replace("libliquid_cache_amd_", "").replace(".so", "")
    dbg = os.environ.get("LC_TIME_SPARSE_DEBUG")
    t_cnt = timed(count_only, None)
    if dbg: print("count ok", t_cnt, flush=True)
    t_ev = timed(ev, 0)
    if dbg: print("eval_hits ok", t_ev, flush=True)
    ctr.zero_()
    ev()
    torch.cuda.synchronize()
    n_hits = int(ctr[0].item())
    if dbg: print("n_hits", n_hits, flush=True)
    t_f = timed(flt, 8)
    if dbg: print("filter ok", t_f, int(ctr[1].item()), flush=True)
    # (timed() left n_hits from its last ev-less loop untouched: counters 0 is still the eval's)
    t_g = timed(gat, 16)
    t_gs = timed(gat_slotted, 16)
    print("%-8s count-only %.2f us  eval_hits(+memset) %.2f us  filter_hits(+memset) %.2f us  gather_bytes_hits(+memset) %.2f us  "
          "slotted %.2f us  hits %d" % (tag, t_cnt, t_ev, t_f, t_g, t_gs, n_hits), flush=True)
    # the PARTITIONED list form (LC_HITS_PARTITIONED): 16 counters on their own cache lines
    P, CS = N.HITS_PARTITIONS, N.HITS_COUNTER_STRIDE
    pctr = torch.zeros(2 * P * CS + 2, dtype=torch.int64, device="cuda")
    q = pctr.data_ptr()
    q2, qb = q + 8 * P * CS, q + 16 * P * CS

    def pzero(ptr, nbytes):
        N.check(cache._lib.lc_device_memset(cache.handle, ptr, 0, nbytes, stream), cache.handle)

    def ptimed(fn, zero_ptr, zero_bytes):
        pctr.zero_()
        scan.eval_hits(like, hits.data_ptr(), cap, q, 0, 0, 0, total.data_ptr(), stream, counters_zeroed=True, partitioned=True)
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            pzero(zero_ptr, zero_bytes)
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3
    tp_ev = ptimed(lambda: scan.eval_hits(like, hits.data_ptr(), cap, q, 0, 0, 0, total.data_ptr(), stream, counters_zeroed=True,
                                          partitioned=True), q, 8 * P * CS)
    pctr.zero_()
    scan.eval_hits(like, hits.data_ptr(), cap, q, 0, 0, 0, total.data_ptr(), stream, counters_zeroed=True, partitioned=True)
    torch.cuda.synchronize()
    n_p = int(pctr[0: P * CS: CS].sum().item())
    tp_f = ptimed(lambda: scan.filter_hits(ne, hits.data_ptr(), q, cap, hits2.data_ptr(), cap, q2, stream, counters_zeroed=True,
                                           partitioned=True), q2, 8 * P * CS)
    tp_g = ptimed(lambda: scan.gather_bytes_hits(hits.data_ptr(), q, cap, views.data_ptr(), data.data_ptr(),
                                                 min(data.numel(), (1 << 31) - 1), qb, 0, stream, counters_zeroed=True,
                                                 partitioned=True), qb, 8)
    tp_gs = ptimed(lambda: scan.gather_bytes_hits(hits.data_ptr(), q, cap, views.data_ptr(), data.data_ptr(),
                                                  min(data.numel(), (1 << 31) - 1), qb, 0, stream, counters_zeroed=True, slotted=True,
                                                  partitioned=True), qb, 8)
    print("%-8s PARTITIONED: eval_hits(+memset) %.2f us  filter_hits(+memset) %.2f us  gather_bytes_hits(+memset) %.2f us  slotted %.2f us  "
          "hits %d" % (tag, tp_ev, tp_f, tp_g, tp_gs, n_p), flush=True)
    scan.close()
    cache.close()


if __name__ == "__main__":
    main()
