#!/bin/bash
# Per-phase cycles of k_str_pred from the timing build (make TIMING=1): for every checkpoint pair the mean cycles per
# entry.  usage: scripts/phase_cycles.sh [bench args]   (run on the GPU box; needs liquid_cache_amd/libliquid_cache_amd_timing.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
export LC_LIB_PATH=$R/liquid_cache_amd/libliquid_cache_amd_timing.so
for t in ${PHASES:-1 2 3 4 5 6 7 8 9}; do
  LC_DEBUG_FLAGS=$((t << 16)) LC_DUMP_COUNTS=/tmp/tm_$t.npy python $R/bench.py --full-line --no-secondary --no-cpu-baseline --no-cold --steps 3 --warmup 1 "$@" > /tmp/tm_$t.log 2>&1 || tail -3 /tmp/tm_$t.log
  python - <<PY
import numpy as np
c = np.load("/tmp/tm_$t.npy").astype(np.int64)
nz = c[c > 0]
import sys
print("histogram (64-wide bins):", np.bincount((c // 64).astype(np.int64))[:12].tolist()) if $t >= 11 else None
print("checkpoint %d: mean %.0f cycles over all entries, %.0f over the %d entries that passed it (max %d)" % ($t, c.mean(), nz.mean() if nz.size else 0, nz.size, c.max()))
PY
done
