#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4e
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4e/pytest_all.txt 2>&1; grep -E "passed|failed|Error|error|assert" gpurun_out/r4e/pytest_all.txt | tail -12
