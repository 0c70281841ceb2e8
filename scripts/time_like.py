"""Kernel time of the headline LIKE scan for a product-library build (LC_LIB_PATH), without any result check — the timing
aid for the phase-stop variants (-DLC_FLAT_STOP / -DLC_LEAN_STOP, whose results are wrong on purpose).
usage: python scripts/time_like.py [--like-path 4] [--rows N] [--needle google] [--iters 30]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--like-path", type=int, default=4)
    ap.add_argument("--rows", type=int, default=99_997_497)
    ap.add_argument("--needle", default="google")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--cold", action="store_true")
    ap.add_argument("--rebuild", action="store_true", help="build the scan-level index a second time in the same process")
    a = ap.parse_args()
    args = bench.parse_args(["--rows", str(a.rows), "--needle", a.needle])
    import torch
    import pyarrow as pa
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_path=a.like_path or None).build()
    n_batches = (args.rows + args.batch_size - 1) // args.batch_size
    ids = bench.stage_url_column(cache, lc, N, args, 0, n_batches, 16)
    scan = cache.scan(ids)
    expr = lc.LiquidExpr.try_new("like", ("%" + a.needle + "%").encode(), pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
    mask = torch.zeros(max(int(scan.mask_words), 1), dtype=torch.int64, device="cuda")
    counts = torch.zeros(max(scan.entries, 1), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    scan.eval(expr, mask.data_ptr(), 0, counts.data_ptr(), stream)
    torch.cuda.synchronize()
    hot = scan.eval_timed(expr, mask.data_ptr(), a.iters, 0, counts.data_ptr(), stream)
    cold = scan.eval_timed_cold(expr, mask.data_ptr(), 5, bench.FLUSH_BYTES, 0, counts.data_ptr(), stream) if a.cold else float("nan")
    built = "%.2f" % scan.info().index_build_ms
    print("%-8s path %d  hot %.2f us  cold %.2f us  index build %s ms  hits %d  %s" % (
        os.path.basename(os.environ.get("LC_LIB_PATH", "default")).replace("libliquid_cache_amd_", "").replace(".so", ""),
        a.like_path, hot * 1e3, cold * 1e3, built, int(counts.sum(dtype=torch.int64).item()), scan.explain(expr)[-110:]), flush=True)
    scan.close()
    if a.rebuild:
        cache.set_option(N.OPT_LIKE_INDEX_CACHE, 0)  # (the first scan's index was kept for adoption: drop that behaviour)
        for k in range(2):
            scan2 = cache.scan(ids)
            scan2.eval(expr, mask.data_ptr(), 0, counts.data_ptr(), stream)
            torch.cuda.synchronize()
            print("         build %d in the same process: %.2f ms" % (k + 2, scan2.info().index_build_ms), flush=True)
            scan2.close()
    cache.close()


if __name__ == "__main__":
    main()
