#!/bin/bash
# usage: scripts/profile_round.sh <round-tag>        (on the GPU box; WORKLOADS="a b" for a partial refresh)
# ONE coherent profile set per round: for EVERY workload of the bench a rocprofv3 kernel trace (stats) and the HBM traffic
# counters (one --pmc pass per counter, no other trace domain), merged into gpurun_out/<tag>/hbm_traffic.json together
# with the measured FETCH_SIZE calibration; the script FAILS if a workload of the list has no figure at the end.
# Copy the summaries into profiles/<tag>/ with scripts/collect_profiles.sh.
tag=${1:-r4}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# one column, back to back (rocprof averages then describe hot launches; the cold figures come from the bench line itself)
B="python $R/bench.py --full-line --scans-per-step 8 --no-secondary --no-needle-classes --no-cpu-baseline --no-cold --rotate 1 --steps 20 --warmup 3"
KREGEX="k_str_pred|k_fixed_pred|k_like_lean|k_like_flat|k_like_scanall|k_fixed_chain|k_fixed_gather|k_sel_entry_counts|k_scan_"
declare -A WL
WL[url_like]="$B --workload url_like"
WL[url_like_k_like_lean]="$B --workload url_like --like-path 3"
WL[url_like_k_str_pred]="$B --workload url_like --like-path 1"
WL[url_like_no_signatures]="$B --workload url_like --no-signatures"
WL[url_like_no_fingerprints]="$B --workload url_like --no-fingerprints"
WL[url_like_1byte]="$B --workload url_like --needle q --needle-ppm 0"
WL[int64_gt_w62]="$B --workload int64_gt --int-bits 62"
WL[int64_gt_w17]="$B --workload int64_gt --int-bits 17 --int-base 1000"
WL[date32_gt_w12]="$B --workload int64_gt --int-kind date32 --int-bits 12 --int-base 8036"
WL[int16_gt_w12]="$B --workload int64_gt --int-kind int16 --int-bits 12 --int-base 0"
WL[decimal_gt_w4]="$B --workload int64_gt --int-kind decimal --int-bits 4 --int-base 0"
WL[tpch_q6]="python $R/bench.py --full-line --scans-per-step 2 --workload tpch_q6 --steps 10 --warmup 2 --no-secondary --no-cpu-baseline"
WL[gather_10pct]="python $R/scripts/gather_profile.py --frac 0.1"
ALL="url_like url_like_k_like_lean url_like_k_str_pred url_like_no_signatures url_like_no_fingerprints url_like_1byte int64_gt_w62 int64_gt_w17 date32_gt_w12 int16_gt_w12 decimal_gt_w4 tpch_q6 gather_10pct"
WORKLOADS=${WORKLOADS:-$ALL}
for wl in $WORKLOADS; do
  cmd=${WL[$wl]}
  [ -z "$cmd" ] && { echo "unknown workload $wl"; exit 2; }
  LC_X=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${wl}_trace" -- $cmd > "$O/${wl}_trace.log" 2>&1
  f=$(find "$O/${wl}_trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/${wl}_kernel_stats.csv"
  grep -h '^{"metric"' "$O/${wl}_trace.log" > "$O/${wl}_bench_line.json"
  extra=""; case $wl in tpch_q6|gather_10pct) ;; *) extra="--steps 3 --warmup 1";; esac
  for c in FETCH_SIZE WRITE_SIZE; do
    LC_X=0 timeout 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$KREGEX" --output-format csv -d "$O/${wl}_$c" -- $cmd $extra > "$O/${wl}_$c.log" 2>&1
  done
done
# FETCH_SIZE calibration on known byte counts (always: the factors are part of the round's record)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_calib" --output-format csv -d "$O/calib_FETCH_SIZE" -- python $R/scripts/pmc_calibrate.py > "$O/calib.log" 2>&1
python $R/scripts/pmc_summary.py "$O" $ALL > "$O/pmc_summary.txt" 2>&1; rc=$?
# the driver-style default run (rotating columns = L3-cold timed loop, cold-primary roofline, needle classes, secondaries)
# AFTER the summary, so that its roofline objects carry this round's traffic figures
mkdir -p "$R/profiles/$tag" && cp "$O/hbm_traffic.json" "$R/profiles/$tag/hbm_traffic.json"
# (the stdout line is the compact record the driver parses; the whole record is bench_detail.json)
[ -z "$NO_FULL" ] && { timeout 1200 python $R/bench.py --steps 20 --warmup 5 --detail-path "$O/bench_detail.json" > "$O/bench_line.json" 2> "$O/full_bench.err"; }
# only the summaries travel back (the raw traces are tens of MB each)
find "$O" -maxdepth 1 -type d \( -name "*_trace" -o -name "*_FETCH_SIZE" -o -name "*_WRITE_SIZE" \) -exec rm -r {} +
tail -30 "$O/pmc_summary.txt"
exit $rc
