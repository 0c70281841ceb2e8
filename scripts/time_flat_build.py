"""k_flat_build alone: build time of the scan-level index of a 100 M-row URL column, synchronous (LC_OPT_LIKE_INDEX_ASYNC = 0:
two workgroups per CU) and on the builder thread (polite: one per CU), three builds each; COUNT(*) checked every time.
usage: python scripts/time_flat_build.py [--rows N]"""
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
    import pyarrow as pa
    import torch
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N
    cache = lc.LiquidCacheBuilder.new().build()
    n_batches = (a.rows + args.batch_size - 1) // args.batch_size
    ids = bench.stage_url_column(cache, lc, N, args, 0, n_batches, 16)
    expr = lc.LiquidExpr.try_new("like", b"%google%", pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
    total = torch.zeros((), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    cache.set_option(N.OPT_LIKE_INDEX_CACHE, 0)  # (a destroyed scan's index is not kept: every scan builds its own)
    cache.set_option(N.OPT_SCAN_CACHE, 0)
    want = None
    for mode in (0, 1):
        cache.set_option(N.OPT_LIKE_INDEX_ASYNC, mode)
        for k in range(3):
            sc = cache.scan(ids)
            sc.eval_count(expr, 0, total.data_ptr(), 0, 0, stream)
            sc.index_wait()
            sc.eval_count(expr, 0, total.data_ptr(), 0, 0, stream)  # (answered by the index that was just built)
            got = int(total.item())
            want = got if want is None else want
            info = sc.info()
            print("%s build %d: %.2f ms, index %.2f GB, COUNT(*) %d (%s), answered by %s" % (
                "builder thread (polite)" if mode else "synchronous", k, info.index_build_ms, info.index_bytes / 1e9, got,
                "ok" if got == want else "MISMATCH", N.LIKE_KERNEL_NAMES.get(int(info.last_like_kernel))), flush=True)
            sc.close()
    cache.close()


if __name__ == "__main__":
    main()
