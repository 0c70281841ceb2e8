#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4d
gcc -std=gnu11 -O1 -Wall -Werror -pthread -I include tests/c_abi/concurrent_callers.c -o /tmp/cc -L liquid_cache_amd -l:libliquid_cache_amd.so -Wl,-rpath,$R/liquid_cache_amd
for m in 1 2 4 8 15; do echo "mode $m"; CC_NO_CHURN=1 CC_MODE=$m timeout 300 /tmp/cc 2>&1 | grep "alone"; done
echo churn; timeout 300 /tmp/cc 2>&1 | tail -n 2
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4d/pytest_all.txt 2>&1; grep -E "passed|failed|Error|error" gpurun_out/r4d/pytest_all.txt | tail -5
