#!/bin/bash
# SQ instruction / cycle counters of the kernels matching a regex, for ANY command (one --pmc pass per counter group, kernel trace
# only).  usage: scripts/pmc_cmd.sh <out-tag> <kernel-regex> <command ...>      env GROUPS_SEL="1 3" picks counter groups
tag=$1; regex=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G[1]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
G[2]="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"
G[3]="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G[4]="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"
G[5]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
for i in ${GROUPS_SEL:-1 2 3 4 5}; do
  (cd $R && timeout 400 rocprofv3 --pmc ${G[$i]} --kernel-trace --kernel-include-regex "$regex" --output-format csv -d $O/g$i -- "$@" > $O/g$i.log 2>&1)
done
python - <<PY
import csv, glob, collections, statistics
acc = collections.defaultdict(list)
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k[0][-40:], k[1], "median per launch %.4g over %d" % (statistics.median(acc[k]), len(acc[k])),
          ("  each: " + " ".join("%.3g" % v for v in acc[k])) if len(acc[k]) <= 8 else "")
PY
rm -rf $O/g*/  # (the raw per-dispatch files are large: gpurun copies at most 64 MiB back)
