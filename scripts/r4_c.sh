#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4c
timeout 600 python bench.py --steps 20 --warmup 5 --secondary-set none > gpurun_out/r4c/bench1.json 2> gpurun_out/r4c/bench1.err; tail -c 1500 gpurun_out/r4c/bench1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4c/bench1.json').read().strip().splitlines()[-1])
r=d['roofline']
print({k:v for k,v in d.items() if k not in ('config','roofline','like_needle_classes','secondary','cpu_baseline_all_cores')})
print(d['config'])
print(r)
PY
bash scripts/scale_dryrun.sh 2 --no-secondary --no-cpu-baseline > gpurun_out/r4c/dry2.json 2> gpurun_out/r4c/dry2.err; tail -c 1500 gpurun_out/r4c/dry2.err; tail -c 3000 gpurun_out/r4c/dry2.json
bash scripts/scale_dryrun.sh 2 --no-secondary --no-cpu-baseline --scaling weak --rows 20000000 > gpurun_out/r4c/dry2w.json 2> gpurun_out/r4c/dry2w.err; tail -c 800 gpurun_out/r4c/dry2w.err; tail -c 1200 gpurun_out/r4c/dry2w.json
