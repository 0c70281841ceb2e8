"""get-with-selection of an Int64 W=62 column at a given selectivity, in a loop — the launches a rocprofv3 pass of
scripts/profile_round.sh sees for the `gather_10pct` workload (k_sel_entry_counts + k_scan_* + k_fixed_gather).
usage: python scripts/gather_profile.py [--frac 0.1] [--iters 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frac", type=float, default=0.1)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rows", type=int, default=99_997_497)
    a = ap.parse_args()
    args = bench.parse_args(["--rows", str(a.rows)])
    args.batch0 = None
    import torch
    import pyarrow as pa
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N
    cache = lc.LiquidCacheBuilder.new().build()
    bits, base = 62, bench.int_base(62)
    ids = bench.stage_int_column(cache, lc, N, args, 1, a.rows, 16, bits=bits, base=base, col=50, kind="int64")
    scan = cache.scan(ids)
    stream = torch.cuda.current_stream().cuda_stream
    words = int(scan.mask_words)
    sel = torch.zeros(max(words, 1), dtype=torch.int64, device="cuda")
    counts = torch.zeros(max(scan.entries, 1), dtype=torch.int32, device="cuda")
    lit = base + int((1 << bits) * (1.0 - a.frac))
    scan.eval(lc.LiquidExpr.try_new(">", lit, pa.int64()), sel.data_ptr(), 0, counts.data_ptr(), stream)
    k = int(counts.sum(dtype=torch.int64).item())
    vals = torch.zeros(max(k, 1) + 8, dtype=torch.int64, device="cuda")
    offs = torch.zeros(scan.entries + 1, dtype=torch.int64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    scan.gather_fixed(vals.data_ptr(), vals.numel() * 8, offs.data_ptr(), sel.data_ptr(), stream)
    e0.record()
    for _ in range(a.iters):
        scan.gather_fixed(vals.data_ptr(), vals.numel() * 8, offs.data_ptr(), sel.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    print('{"metric": "gather", "selected_rows": %d, "rows": %d, "ms": %.5f}' % (k, a.rows, e0.elapsed_time(e1) / a.iters), flush=True)
    scan.close()
    cache.close()


if __name__ == "__main__":
    main()
