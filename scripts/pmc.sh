#!/bin/bash
# usage: scripts/pmc.sh <tag> <bench args...> ; one rocprofv3 --pmc pass per counter group (no trace domains mixed in)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "FETCH_SIZE WRITE_SIZE"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}/$name -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $R/gpurun_out/pmc_${tag}/$name.log 2>&1
done
python $R/scripts/pmc_summary.py $R/gpurun_out/pmc_${tag}
