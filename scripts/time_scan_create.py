"""Host wall clock of the per-query steps of a host that creates a scan per query: lc_scan_create over the 12,207 entries of
the bench's URL column, the first evaluation on it (records, automata, adopted index, one launch), lc_scan_destroy.
usage: python scripts/time_scan_create.py [--rows N]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=99_997_497)
    a = ap.parse_args()
    args = bench.parse_args(["--rows", str(a.rows)])
    import torch
    import pyarrow as pa
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N
    cache = lc.LiquidCacheBuilder.new().build()
    n_batches = (args.rows + args.batch_size - 1) // args.batch_size
    ids = bench.stage_url_column(cache, lc, N, args, 0, n_batches, 16)
    expr = lc.LiquidExpr.try_new("like", b"%google%", pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
    stream = torch.cuda.current_stream().cuda_stream
    # the C call alone (the Python mirror converts 12,207 id objects first)
    import ctypes as C
    import numpy as np
    ids_np = np.ascontiguousarray(np.asarray([int(e) for e in ids], dtype=np.uint64))
    for it in range(4):
        h = C.c_void_p()
        t0 = time.perf_counter()
        N.check(cache._lib.lc_scan_create(cache.handle, len(ids_np), ids_np.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(h)), cache.handle)
        t1 = time.perf_counter()
        cache._lib.lc_scan_destroy(h)
        t2 = time.perf_counter()
        print("C call %d: lc_scan_create %.0f us, lc_scan_destroy %.0f us" % (it, (t1 - t0) * 1e6, (t2 - t1) * 1e6), flush=True)
    mask = None
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scan = cache.scan(ids)
        t1 = time.perf_counter()
        if mask is None:
            mask = torch.zeros(max(int(scan.mask_words), 1), dtype=torch.int64, device="cuda")
        scan.eval(expr, mask.data_ptr(), 0, 0, stream)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        scan.eval(expr, mask.data_ptr(), 0, 0, stream)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        scan.close()
        t4 = time.perf_counter()
        print("query %d: scan create %.0f us, first evaluation %.0f us, second %.0f us, destroy %.0f us" % (
            it, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6), flush=True)
    cache.close()


if __name__ == "__main__":
    main()
