mkdir -p gpurun_out/r3e
timeout 300 python -m pytest tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -8
for p in 1 2 3 0; do
  timeout 200 python bench.py --full-line --no-secondary --no-cpu-baseline --rotate 1 --steps 20 --warmup 3 --like-path $p > gpurun_out/r3e/path$p.json 2> gpurun_out/r3e/path$p.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r3e/path$p.json').read().strip().splitlines()[-1])
r=d['roofline']
print('path $p', d['config'].get('evaluation_path'), 'hot us %.2f cold us %.2f step us %.2f hits %d bytes %d'%(r['kernel_ms_hot']*1e3, r['kernel_ms']*1e3, d['ms_per_step']*1e3, d['config']['hits'], r['kernel_bytes_per_launch']))
PY
done
