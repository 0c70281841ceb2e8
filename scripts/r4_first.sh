#!/bin/bash
# round 4, first GPU call: parity of the scan-level index (k_like_flat) + A/B against k_like_lean
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4a
cd $R
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -15
echo "== lean (path 3)"
bash scripts/ab_variants.sh "default" --like-path 3
echo "== flat (path 0), waves per workgroup 1 / 2 / 4"
bash scripts/ab_variants.sh "default fw2 fw4" --like-path 0
cp gpurun_out/ab/default.json gpurun_out/r4a/flat_default.json
