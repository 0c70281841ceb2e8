#!/bin/bash
# The width-specialised register kernels (k_fixed_pred_reg_w) against the mixed-width kernel (-DLC_X_UNIFORM_W=0 build in
# liquid_cache_amd/variants/libliquid_cache_amd_uw0.so): hot / L3-cold kernel time of the narrow-integer workloads, 100 M rows.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
declare -A WL
WL[date32_gt_w12]="--workload int64_gt --int-kind date32 --int-bits 12 --int-base 8036"
WL[int16_gt_w12]="--workload int64_gt --int-kind int16 --int-bits 12 --int-base 0"
WL[decimal_gt_w4]="--workload int64_gt --int-kind decimal --int-bits 4 --int-base 0"
WL[int64_gt_w13]="--workload int64_gt --int-bits 13 --int-base 1000"
WL[int64_gt_w17]="--workload int64_gt --int-bits 17 --int-base 1000"
for wl in date32_gt_w12 int16_gt_w12 decimal_gt_w4 int64_gt_w13 int64_gt_w17; do
  for lib in default uw0; do
    if [ $lib = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$lib.so; fi
    python bench.py ${WL[$wl]} --no-secondary --no-cpu-baseline --steps 5 --warmup 2 --full-line 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-14s %-8s kernel %s  hot %.2f us  cold %.2f us  (hits %s)' % ('$wl', '$lib', r['kernel'], (r.get('kernel_ms_hot') or 0)*1e3, (r.get('kernel_ms_l3_cold') or r['kernel_ms'])*1e3, d['config'].get('hits')))"
  done
done
