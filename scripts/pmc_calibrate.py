"""Launch the counter-calibration kernels (lc_calibrate_read): run under
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex k_calib -d <dir> -- python scripts/pmc_calibrate.py
Each shape reads exactly BYTES bytes per launch; scripts/pmc_summary.py divides by the counter to get bytes per count."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (initialises the HIP runtime the way bench.py does)
import liquid_cache_amd as lc  # noqa: E402
from liquid_cache_amd import _native as N  # noqa: E402

BYTES = 1 << 30
cache = lc.LiquidCacheBuilder.new().build()
L = N.load()
for shape in (4, 8, 16, 1008):
    N.check(N.load_bench().lc_calibrate_read(cache.handle, BYTES, shape, 5), cache.handle)
print("calibration launches done: %d bytes per launch" % BYTES)
cache.close()
