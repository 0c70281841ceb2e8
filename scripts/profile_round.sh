#!/bin/bash
# usage: scripts/profile_round.sh <round-tag>
# rocprofv3 kernel-trace stats + HBM traffic counters (one --pmc pass per counter, no other trace domains) for both
# bench workloads.  Outputs land in gpurun_out/<tag>/; copy the summaries into profiles/<tag>/.
tag=${1:-r1}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in url_like int64_gt; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${wl}_trace -- python $R/bench.py --workload $wl --no-q21 --steps 20 --warmup 3 > $O/${wl}_trace.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_str_pred|k_fixed_pred" --output-format csv -d $O/${wl}_$c -- python $R/bench.py --workload $wl --no-cpu-baseline --no-q21 --steps 3 --warmup 1 > $O/${wl}_$c.log 2>&1
  done
done
# the same LIKE scan with the reference's own prefilter only (no bigram signature index staged)
LC_NO_SIGNATURES=1 timeout 300 python $R/bench.py --workload url_like --steps 5 --warmup 1 --no-cpu-baseline --no-q21 > $O/url_like_no_signatures.log 2>&1
python $R/scripts/pmc_summary.py $O
grep -h '^{"metric"' $O/url_like_no_signatures.log | cut -c1-2000
for wl in url_like int64_gt; do grep -h '^{"metric"' $O/${wl}_trace.log; f=$(find $O/${wl}_trace -name "*kernel_stats.csv" | head -1); head -4 $f; done
