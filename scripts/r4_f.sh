#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4f/pytest_all.txt 2>&1; grep -E "passed|failed|Error|error" gpurun_out/r4f/pytest_all.txt | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for rows in 25000000 50000000 75000000 99997497; do timeout 120 python scripts/time_like.py --like-path 0 --rows $rows --cold 2>&1 | tail -1; done | tee gpurun_out/r4f/row_sweep.txt
timeout 300 python bench.py --steps 20 --warmup 5 --secondary-set none --no-cpu-baseline > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err; tail -c 300 gpurun_out/r4f/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r4f/bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_hot'], r['traffic'], r['frac'])"
