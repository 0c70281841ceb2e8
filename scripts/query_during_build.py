"""What a query costs WHILE the builder thread builds another table's scan-level index: table A has its index, table B's first
LIKE starts B's build, and COUNT(*) queries on A are timed one after the other until B's build is done (host wall clock per
query, synchronised on the query's stream).  usage: python scripts/query_during_build.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import copy
    import pyarrow as pa
    import torch
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N
    args = bench.parse_args([])
    cache = lc.LiquidCacheBuilder.new().build()
    n_batches = (args.rows + args.batch_size - 1) // args.batch_size
    expr = lc.LiquidExpr.try_new("like", b"%google%", pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
    total = torch.zeros((), dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    stream = side.cuda_stream
    tables = []
    for t in range(4):
        a2 = copy.copy(args)
        a2.seed = args.seed + 7919 * t
        tables.append(bench.stage_url_column(cache, lc, N, a2, 0, n_batches, 16, file_id=400 + t))
    scan_a = cache.scan(tables[0])
    scan_a.eval_count(expr, 0, total.data_ptr(), 0, 0, stream)
    scan_a.index_wait()
    for _ in range(20):
        scan_a.eval_count(expr, 0, total.data_ptr(), 0, 0, stream)
    side.synchronize()
    quiet = []
    for _ in range(200):
        t0 = time.perf_counter()
        scan_a.eval_count(expr, 0, total.data_ptr(), 0, 0, stream)
        side.synchronize()
        quiet.append((time.perf_counter() - t0) * 1e6)
    for t in (1, 2, 3):
        scan_b = cache.scan(tables[t])
        scan_b.eval_count(expr, 0, total.data_ptr(), 0, 0, stream)  # (starts B's build on the builder thread)
        side.synchronize()
        busy = []
        t_all = time.perf_counter()
        # (the window: the build's own kernel time is 3-5 ms and its 2 GB allocation comes first; `index_build_pending` only
        # clears at scan_b's next evaluation, so the clock decides)
        while time.perf_counter() - t_all < 0.008:
            t0 = time.perf_counter()
            scan_a.eval_count(expr, 0, total.data_ptr(), 0, 0, stream)
            side.synchronize()
            busy.append((time.perf_counter() - t0) * 1e6)
        build_ms = (time.perf_counter() - t_all) * 1e3
        scan_b.index_wait()
        print("table %d: window %.2f ms (build kernel %.2f ms); %d queries on table A in it: median %.1f us, max %.1f us"
              "   (quiet device: median %.1f us, max %.1f us)" % (t, build_ms, scan_b.info().index_build_ms, len(busy),
                                                                 float(np.median(busy)) if busy else float("nan"),
                                                                 max(busy) if busy else float("nan"), float(np.median(quiet)), max(quiet)), flush=True)
        scan_b.close()
    scan_a.close()
    cache.close()


if __name__ == "__main__":
    main()
