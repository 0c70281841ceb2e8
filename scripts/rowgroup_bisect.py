"""Host time per row-group call (lc_bench_rowgroup_run, one thread) under option settings — a bisect aid for what a call pays."""
import os
import sys
import ctypes as C

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    args = bench.parse_args([])
    import pyarrow as pa
    import torch
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N
    B = N.load_bench()
    expr = lc.LiquidExpr.try_new("like", b"%google%", pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
    pred = expr.as_predicate()
    n_batches = (args.rows + args.batch_size - 1) // args.batch_size
    only = os.environ.get("RG_ONLY")
    for label, opts in (("async=1 cache=32", {N.OPT_LIKE_INDEX_ASYNC: 1, N.OPT_SCAN_CACHE: 32}),
                        ("async=0 cache=32", {N.OPT_LIKE_INDEX_ASYNC: 0, N.OPT_SCAN_CACHE: 32}),
                        ("async=1 cache=0", {N.OPT_LIKE_INDEX_ASYNC: 1, N.OPT_SCAN_CACHE: 0}),
                        ("async=0 cache=0", {N.OPT_LIKE_INDEX_ASYNC: 0, N.OPT_SCAN_CACHE: 0})):
        if only and only != label:
            continue
        b = lc.LiquidCacheBuilder.new()
        for k, v in opts.items():
            b = b.with_option(k, v)
        cache = b.build()
        ids = bench.stage_url_column(cache, lc, N, args, 0, n_batches, 16)
        ids_np = np.ascontiguousarray(np.asarray([int(e) for e in ids], dtype=np.uint64))
        rgb = args.row_group_batches
        begins = list(range(0, len(ids), rgb)) + [len(ids)]
        gb = np.ascontiguousarray(np.asarray(begins, dtype=np.uint64))
        for rep in range(2):
            for threads, gps in ((1, 1), (8, 1), (1, len(begins) - 1)):
                st = N.RowGroupStats()
                rc = B.lc_bench_rowgroup_run(cache._ctx, len(begins) - 1, gb.ctypes.data_as(C.POINTER(C.c_uint64)),
                                             ids_np.ctypes.data_as(C.POINTER(C.c_uint64)), C.cast(C.byref(pred), C.c_void_p), threads, 5, 0, gps,
                                             C.byref(st))
                print("%-18s rep %d threads %d groups/call %3d: rc %d call_us_host %.2f pass_us %.1f first_pass_ms %.1f hits %d" % (
                    label, rep, threads, gps, rc, st.call_us_mean, st.wall_s / max(st.passes, 1) * 1e6, st.first_pass_s * 1e3, st.hits), flush=True)
        cache.close()


if __name__ == "__main__":
    main()
