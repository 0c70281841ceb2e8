#!/bin/bash
# usage: scripts/pmc1.sh <tag> "<counters>" [bench args]; ONE rocprofv3 --pmc pass restricted to the scan kernels
tag=$1; ctrs=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_$tag
timeout 240 rocprofv3 --pmc $ctrs --kernel-trace --kernel-include-regex "k_str_pred|k_fixed_pred" --output-format csv -d $R/gpurun_out/pmc_$tag/run -- python $R/bench.py --full-line --no-cpu-baseline --no-q21 --steps 3 --warmup 1 "$@" > $R/gpurun_out/pmc_$tag/log.txt 2>&1
python $R/scripts/pmc_summary.py $R/gpurun_out/pmc_$tag
