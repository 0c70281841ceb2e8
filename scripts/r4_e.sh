#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4e
( time timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary-set micro,sweep > gpurun_out/r4e/bench_micro.json 2> gpurun_out/r4e/bench_micro.err ) 2>&1 | grep real; tail -c 800 gpurun_out/r4e/bench_micro.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4e/bench_micro.json').read().strip().splitlines()[-1])
sec=d['secondary']
for k,v in sec.get('micro',{}).items():
    if 'error' in v: print(k, v); continue
    if 'frac' in v: print('%-34s frac %.3f hot %.3f  ms %.4f hot %.4f sel %.4f' % (k, v['frac'], v.get('frac_hot',0), v['kernel_ms'], v.get('kernel_ms_hot',0), v.get('selectivity',-1)))
    else: print(k, v)
sw=sec.get('clickbench_pushdown_sweep',{})
print({k:v for k,v in sw.items() if k!='queries'})
for q,v in sw.get('queries',{}).items(): print(q, 'ms %.3f frac %.3f passes %d out %d  %s'%(v['ms'], v['frac'], v['passes'], v['rows_out'], v['filter'][:70]))
PY
