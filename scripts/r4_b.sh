#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -x -q > gpurun_out/r4b/pytest.txt 2>&1; grep -E "passed|failed" gpurun_out/r4b/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary > gpurun_out/r4b/full_bench.json 2> gpurun_out/r4b/full_bench.err; tail -c 600 gpurun_out/r4b/full_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4b/full_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value %.3e rows/s step %.2f us kernel cold %.2f hot %.2f own bytes %d frac %.3f path %s'%(d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3, r.get('kernel_ms_hot',0)*1e3, r['kernel_bytes_per_launch'], r['frac'], d['config']['evaluation_path']))
for k,v in d.get('like_needle_classes',{}).items(): print(k, v['kernel_us_hot'], v['kernel_us_l3_cold'], v['mask_equals_cpu_oracle'], v['path'])
PY
