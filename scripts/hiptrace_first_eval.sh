#!/bin/bash
# HIP API calls around the first k_like_lean launch of scripts/first_eval_profile.py (rocprofv3 --hip-trace --kernel-trace; no counters)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/hiptrace -- python $R/scripts/first_eval_profile.py > $O/hiptrace.log 2>&1
cd $R
grep -- "run 0" $O/hiptrace.log
python - <<PY
import csv,glob
f=glob.glob("$O/hiptrace/**/*hip_api_trace.csv",recursive=True)
k=glob.glob("$O/hiptrace/**/*kernel_trace.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
ks=list(csv.DictReader(open(k[0])))
lean=[r for r in ks if "k_like_lean" in r["Kernel_Name"]]
t0=int(lean[0]["Start_Timestamp"]); t1=int(lean[0]["End_Timestamp"])
print("first k_like_lean: dur us", (t1-t0)/1e3)
for r in ks:
    s=int(r["Start_Timestamp"])
    if t0-600000 < s < t1+1500000: print("KERNEL %9.1f %8.1f us %s" % ((s-t0)/1e3,(int(r["End_Timestamp"])-s)/1e3, r["Kernel_Name"][:60]))
win=[r for r in rows if int(r["Start_Timestamp"])>t0-600000 and int(r["Start_Timestamp"])<t1+1500000]
for r in win:
    print("%9.1f %8.1f us  tid %s  %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Thread_Id"), r["Function"]))
PY
