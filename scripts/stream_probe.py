"""Measured streaming-read ceiling of the device (lc_probe_stream_read): time and GB/s of a read-only kernel for the byte
counts of the scan kernels, hot and L3-cold, over a few grid sizes.  Writes one JSON document to stdout."""
import ctypes as C
import json
import sys

import liquid_cache_amd as lc
from liquid_cache_amd import _native as N


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [39_000_000, 80_000_000, 163_000_000, 226_000_000, 775_000_000, 1_500_000_000]
    cache = lc.LiquidCacheBuilder.new().build()
    B = N.load_bench()
    out = []
    try:
        for b in sizes:
            best = None
            for grid in (1024, 2048, 4096, 8192):
                hot, cold = C.c_double(), C.c_double()
                rc = B.lc_probe_stream_read(cache._ctx, b, 20, grid, C.byref(hot), C.byref(cold))
                if rc != 0:
                    raise RuntimeError("lc_probe_stream_read: %d" % rc)
                r = {"bytes": b, "grid": grid, "hot_us": round(hot.value, 2), "cold_us": round(cold.value, 2),
                     "hot_gbs": round(b / hot.value / 1e3, 1), "cold_gbs": round(b / cold.value / 1e3, 1)}
                out.append(r)
                if best is None or r["cold_us"] < best["cold_us"]:
                    best = r
            print("# %d bytes: best cold %.1f us (%.0f GB/s, grid %d), hot %.1f us (%.0f GB/s)" % (
                b, best["cold_us"], best["cold_gbs"], best["grid"], best["hot_us"], best["hot_gbs"]), file=sys.stderr)
    finally:
        cache.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
