"""lc_eval_predicate_row_groups (mode 0) and lc_scan_eval_count_groups (mode 1) from T threads, each over its share of the row
groups, through the bench driver — on its own, for tracing.  usage: python scripts/many_call_threads.py [threads] [mode]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    modes = [int(sys.argv[2])] if len(sys.argv) > 2 else [0, 1]
    import pyarrow as pa
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N
    args = bench.parse_args([])
    cache = lc.LiquidCacheBuilder.new().build()
    n_batches = (args.rows + args.batch_size - 1) // args.batch_size
    ids = bench.stage_url_column(cache, lc, N, args, 0, n_batches, 16)
    expr = lc.LiquidExpr.try_new("like", b"%google%", pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
    pred = expr.as_predicate()
    B = N.load_bench()
    rgb = args.row_group_batches
    ids_np = np.ascontiguousarray(np.asarray([int(e) for e in ids], dtype=np.uint64))
    begins = list(range(0, len(ids), rgb)) + [len(ids)]
    gb = np.ascontiguousarray(np.asarray(begins, dtype=np.uint64))
    B.lc_bench_rowgroup_many.restype = C.c_int32
    B.lc_bench_rowgroup_many.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_int32, C.POINTER(N.RowGroupStats)]
    for mode in modes:
        for rep in range(2):
            st = N.RowGroupStats()
            rc = B.lc_bench_rowgroup_many(cache._ctx, len(begins) - 1, gb.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          ids_np.ctypes.data_as(C.POINTER(C.c_uint64)), C.cast(C.byref(pred), C.c_void_p), threads, 20, mode,
                                          C.byref(st))
            print("mode %d threads %d rep %d: rc %d, pass %.1f us, call %.1f us, first pass %.2f ms, hits %d" % (
                mode, threads, rep, rc, st.wall_s / max(st.passes, 1) * 1e6, st.call_us_mean, st.first_pass_s * 1e3, st.hits), flush=True)
    cache.close()


if __name__ == "__main__":
    main()
