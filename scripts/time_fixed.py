"""Kernel time of `col > literal` over the narrow integer columns of bench.py's secondary set for a product-library build
(LC_LIB_PATH), without any result check — the timing aid for -DLC_X_NOSHIFT / -DLC_X_NOPARK builds (wrong results on purpose).
usage: python scripts/time_fixed.py [--rows N] [--iters 30]"""
import argparse
import datetime
import decimal
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
    stream = torch.cuda.current_stream().cuda_stream
    tag = os.path.basename(os.environ.get("LC_LIB_PATH", "default")).replace("libliquid_cache_amd_", "").replace(".so", "")
    specs = [("date32_gt_w12", "date32", 12, 8036, pa.date32(), 51), ("int16_gt_w12", "int16", 12, 0, pa.int16(), 52),
             ("decimal_gt_w4", "decimal", 4, 0, pa.decimal128(15, 2), 53), ("int64_gt_w17", "int64", 17, 1000, pa.int64(), 54)]
    for name, kind, bits, base, dtype, col in specs:
        ids = bench.stage_int_column(cache, lc, N, args, 1, a.rows, 16, bits=bits, base=base, col=col, kind=kind)
        scan = cache.scan(ids)
        lit = base + (1 << (bits - 1))
        lit_v = decimal.Decimal(lit) / 100 if kind == "decimal" else (
            datetime.date(1970, 1, 1) + datetime.timedelta(days=lit) if kind == "date32" else lit)
        expr = lc.LiquidExpr.try_new(">", lit_v, dtype)
        mask = torch.zeros(max(int(scan.mask_words), 1), dtype=torch.int64, device="cuda")
        counts = torch.zeros(max(scan.entries, 1), dtype=torch.int32, device="cuda")
        scan.eval(expr, mask.data_ptr(), 0, counts.data_ptr(), stream)
        torch.cuda.synchronize()
        hot = scan.eval_timed(expr, mask.data_ptr(), a.iters, 0, counts.data_ptr(), stream)
        cold = scan.eval_timed_cold(expr, mask.data_ptr(), 8, bench.FLUSH_BYTES, 0, counts.data_ptr(), stream)
        print("%-8s %-14s hot %.2f us  cold %.2f us  hits %d" % (tag, name, hot * 1e3, cold * 1e3, int(counts.sum(dtype=torch.int64).item())),
              flush=True)
        scan.close()
        cache.evict(ids)
    cache.close()


if __name__ == "__main__":
    main()
