#!/bin/bash
# A/B of product-library variants (make VARIANT=...): usage scripts/ab_variants.sh "<variant names>" [bench args]
# prints hot / cold kernel time and the step time of the headline scan for every variant ("default" = the shipped file)
R=${GRAFT_REPO_ROOT:-/root/repo}
names=$1; shift
mkdir -p $R/gpurun_out/ab
for v in $names; do
  if [ "$v" = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$v.so; fi
  timeout 200 python $R/bench.py --full-line --no-secondary --no-cpu-baseline --rotate 1 --steps 20 --warmup 3 "$@" > $R/gpurun_out/ab/$v.json 2> $R/gpurun_out/ab/$v.err
  python - <<PY
import json
try:
    d=json.loads(open('$R/gpurun_out/ab/$v.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-10s %-40s hot %.2f us cold %.2f us step %.2f us hits %d'%('$v', d['config'].get('evaluation_path','')[:40], r.get('kernel_ms_hot',0)*1e3, r['kernel_ms']*1e3, d['ms_per_step']*1e3, d['config']['hits']))
except Exception as e:
    print('$v failed', e); print(open('$R/gpurun_out/ab/$v.err').read()[-600:])
PY
done
