#!/bin/bash
# Instruction-issue counters of the string predicate kernel (one --pmc pass per group, kernel trace only).
# usage: scripts/pmc_sq.sh <out-tag> [bench args]      env GROUPS_SEL="1 2" picks counter groups
tag=${1:-sq}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --full-line --no-secondary --no-cold --no-cpu-baseline --rotate 1 --steps 3 --warmup 1"
G[1]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
G[2]="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"
G[3]="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G[4]="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"
G[5]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
for i in ${GROUPS_SEL:-1 2 3 4 5}; do
  timeout 240 rocprofv3 --pmc ${G[$i]} --kernel-trace --kernel-include-regex "k_str_pred|k_fixed_pred|k_fixed_chain|k_like" --output-format csv -d $O/g$i -- $B "$@" > $O/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
# the timed launches are the most frequent value pattern: report the median
import statistics
for k in sorted(acc): print(k[0][-22:], k[1], "median per launch %.4g over %d" % (statistics.median(acc[k]), len(acc[k])))
PY
