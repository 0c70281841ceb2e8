"""Kernel time of string `=` / `<>` over the bench's URL column, through the scan-level index and through k_str_pred.
usage: python scripts/time_str_eq.py [--rows N]"""
import argparse
import os
import sys

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
    for path in (0, 1):
        cache = lc.LiquidCacheBuilder.new().with_index_options(like_path=path or None).build()
        n_batches = (args.rows + args.batch_size - 1) // args.batch_size
        ids = bench.stage_url_column(cache, lc, N, args, 0, n_batches, 16)
        scan = cache.scan(ids)
        value = cache.get(ids[5]).with_selection([i == 17 for i in range(int(scan.rows_of(5) if hasattr(scan, "rows_of") else 8192))]).read()[0].as_py()
        mask = torch.zeros(max(int(scan.mask_words), 1), dtype=torch.int64, device="cuda")
        counts = torch.zeros(max(scan.entries, 1), dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for op in ("=", "!="):
            expr = lc.LiquidExpr.try_new(op, value, pa.string())
            scan.eval(expr, mask.data_ptr(), 0, counts.data_ptr(), stream)
            torch.cuda.synchronize()
            hot = scan.eval_timed(expr, mask.data_ptr(), 20, 0, counts.data_ptr(), stream)
            cold = scan.eval_timed_cold(expr, mask.data_ptr(), 5, bench.FLUSH_BYTES, 0, counts.data_ptr(), stream)
            alg, own = scan.traffic_model(expr, False)
            print("path %d  %-2s %-40r hot %.1f us cold %.1f us hits %d own %.1f MB  %s" % (
                path, op, value[:40], hot * 1e3, cold * 1e3, int(counts.sum(dtype=torch.int64).item()), own / 1e6,
                scan.explain(expr)[:70]), flush=True)
        scan.close()
        cache.close()


if __name__ == "__main__":
    main()
