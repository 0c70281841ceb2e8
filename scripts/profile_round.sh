#!/bin/bash
# usage: scripts/profile_round.sh <round-tag>
# rocprofv3 kernel-trace stats + HBM traffic counters (one --pmc pass per counter, no other trace domains) for the bench
# workloads.  Outputs land in gpurun_out/<tag>/; copy the summaries into profiles/<tag>/ (scripts/collect_profiles.sh).
tag=${1:-r3}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# one column, back to back (rocprof averages then describe hot launches; the cold figures come from the bench line itself)
B="python $R/bench.py --no-secondary --no-needle-classes --no-cpu-baseline --no-cold --rotate 1 --steps 20 --warmup 3"
declare -A WL
WL[url_like]="--workload url_like"
WL[url_like_k_str_pred]="--workload url_like --like-path 1"
WL[url_like_no_fingerprints]="--workload url_like --no-fingerprints"
WL[int64_gt_w62]="--workload int64_gt --int-bits 62"
WL[date32_gt_w12]="--workload int64_gt --int-kind date32 --int-bits 12 --int-base 8036"
WL[int16_gt_w12]="--workload int64_gt --int-kind int16 --int-bits 12 --int-base 0"
WL[decimal_gt_w4]="--workload int64_gt --int-kind decimal --int-bits 4 --int-base 0"
run_one() {  # name, env prefix, args
  local wl=$1 envp=$2 a=$3
  env $envp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${wl}_trace -- $B $a > $O/${wl}_trace.log 2>&1
  f=$(find $O/${wl}_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${wl}_kernel_stats.csv
  grep -h '^{"metric"' $O/${wl}_trace.log > $O/${wl}_bench_line.json
  for c in FETCH_SIZE WRITE_SIZE; do
    env $envp timeout 240 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_str_pred|k_fixed_pred|k_like_lean|k_fixed_chain" --output-format csv -d $O/${wl}_$c -- $B $a --steps 3 --warmup 1 > $O/${wl}_$c.log 2>&1
  done
}
# WORKLOADS="url_like date32_gt_w12" scripts/profile_round.sh <tag>: only these (a partial refresh after a kernel change)
ALL="full url_like url_like_k_str_pred url_like_no_fingerprints int64_gt_w62 date32_gt_w12 int16_gt_w12 decimal_gt_w4 tpch_q6 url_like_no_signatures calib"
WORKLOADS=${WORKLOADS:-$ALL}
has() { [[ " $WORKLOADS " == *" $1 "* ]]; }
for wl in url_like url_like_k_str_pred url_like_no_fingerprints int64_gt_w62 date32_gt_w12 int16_gt_w12 decimal_gt_w4; do has $wl && run_one $wl "LC_X=0" "${WL[$wl]}"; done
# BASELINE.json config 4 at its full size: the Q6-shaped chain over 600,037,902 rows (kernel trace only)
if has tpch_q6; then
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tpch_q6_trace -- python $R/bench.py --workload tpch_q6 --steps 10 --warmup 2 > $O/tpch_q6_trace.log 2>&1
f=$(find $O/tpch_q6_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/tpch_q6_kernel_stats.csv
grep -h '^{"metric"' $O/tpch_q6_trace.log > $O/tpch_q6_bench_line.json
fi
# the same LIKE scan with the reference's own prefilter only (no bigram signature index staged)
has url_like_no_signatures && run_one url_like_no_signatures "LC_X=0" "--workload url_like --no-signatures"
# FETCH_SIZE calibration on known byte counts
has calib && timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_calib" --output-format csv -d $O/calib_FETCH_SIZE -- python $R/scripts/pmc_calibrate.py > $O/calib.log 2>&1
python $R/scripts/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
# the driver-style default run (rotating columns = L3-cold timed loop, cold-primary roofline, needle classes, secondaries)
has full && { timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/full_bench_line.json 2> $O/full_bench.err; }
# only the summaries travel back (the raw traces are tens of MB)
rm -rf $O/*_trace $O/*_FETCH_SIZE $O/*_WRITE_SIZE
for wl in url_like url_like_k_str_pred url_like_no_signatures url_like_no_fingerprints int64_gt_w62 date32_gt_w12 int16_gt_w12 decimal_gt_w4; do has $wl && { echo "== $wl"; grep -v "build_signatures\|copyBuffer\|fillBuffer\|at::native" $O/${wl}_kernel_stats.csv | head -4 | cut -c1-200; }; done
tail -40 $O/pmc_summary.txt
