#!/bin/bash
# A/B of the fixed-width predicate kernels over the five integer workloads: variants built with `make VARIANT=... EXTRA=...`
# usage: scripts/ab_int2.sh <out-tag> [variant ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; tag=$1; shift; mkdir -p gpurun_out/$tag
for v in default "$@"; do
  if [ "$v" = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$v.so; fi
  python scripts/time_int.py 2>&1 | grep -E "cold|error" | tee -a gpurun_out/$tag/ab_int.txt
done
