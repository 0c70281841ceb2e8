"""Hot / L3-cold kernel time of the integer predicate workloads (bench.secondary_int_columns) for a product-library build
(LC_LIB_PATH) — the timing aid of scripts/ab_int2.sh; results are printed, not checked (ablation builds are wrong on purpose).
usage: python scripts/time_int.py [--rows N] [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=99_997_497)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    args = bench.parse_args(["--rows", str(a.rows)])
    import torch
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N
    cache = lc.LiquidCacheBuilder.new().build()
    stream = torch.cuda.current_stream().cuda_stream
    out = bench.secondary_int_columns(cache, lc, N, args, a.rows, 16, torch, stream, a.iters)
    tag = os.path.basename(os.environ.get("LC_LIB_PATH", "default")).replace("libliquid_cache_amd_", "").replace(".so", "")
    for k, v in out.items():
        if "kernel_ms" in v:
            print("%-8s %-16s cold %6.1f hot %6.1f us  probe cold %6.1f hot %6.1f  hits %d" % (
                tag, k, v["kernel_ms"] * 1e3, v["kernel_ms_hot"] * 1e3, v.get("read_probe", {}).get("cold_us", 0),
                v.get("read_probe", {}).get("hot_us", 0), v.get("hits", -1)), flush=True)
        else:
            print(tag, k, v, flush=True)
    cache.close()


if __name__ == "__main__":
    main()
