#!/bin/bash
# usage: scripts/collect_profiles.sh <round-tag>
# Copies the summaries scripts/profile_round.sh left under gpurun_out/<tag>/ into profiles/<tag>/ (tracked).
tag=${1:-r4}
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/$tag
D=$R/profiles/$tag
mkdir -p $D
cp $S/*_kernel_stats.csv $S/*_bench_line.json $S/bench_line.json $S/bench_detail.json $D/ 2>/dev/null
[ -f $S/hbm_traffic.json ] && cp $S/hbm_traffic.json $D/
[ -f $S/pmc_summary.txt ] && cp $S/pmc_summary.txt $D/hbm_traffic_pmc.txt
ls -la $D
