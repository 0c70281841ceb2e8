#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4g
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4g/pytest.txt 2>&1; grep -E "passed|failed|Error|error|assert" gpurun_out/r4g/pytest.txt | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
