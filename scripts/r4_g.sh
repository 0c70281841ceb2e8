#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4g
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4g/pytest.txt 2>&1; grep -E "passed|failed|Error|error|assert" gpurun_out/r4g/pytest.txt | tail -4
timeout 300 python bench.py --steps 10 --warmup 3 --no-fingerprints --no-secondary --rotate 1 > gpurun_out/r4g/nofp.json 2> gpurun_out/r4g/nofp.err; tail -c 300 gpurun_out/r4g/nofp.err
python -c "
import json; d=json.loads(open('gpurun_out/r4g/nofp.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['config']['evaluation_path'][:80]); print('hot %.1f cold %.1f own %d frac %.3f hits %d cpu %s'%(r['kernel_ms_hot']*1e3, r['kernel_ms']*1e3, r['kernel_bytes_per_launch'], r['frac'], d['config']['hits'], d['config'].get('hits_match_cpu_oracle')))"
