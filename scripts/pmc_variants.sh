#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of k_like_lean for product-library variants: scripts/pmc_variants.sh "<variants>" [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
names=$1; shift
O=$R/gpurun_out/pmcv; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in $names; do
  if [ "$v" = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_like_lean" --output-format csv -d $O/${v}_$c -- python $R/bench.py --full-line --no-secondary --no-cpu-baseline --no-cold --rotate 1 --steps 3 --warmup 1 "$@" > $O/${v}_$c.log 2>&1
  done
  python - <<PY
import csv, glob, statistics
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob("$O/${v}_%s/**/*counter_collection.csv" % c, recursive=True):
        vals += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_like_lean" in r["Kernel_Name"]]
    out[c] = statistics.median(vals) if vals else float("nan")
print("%-8s fetch %.1f MB (x2 corrected) write %.1f MB  launches %d" % ("$v", out["FETCH_SIZE"] * 1024 * 2 / 1e6, out["WRITE_SIZE"] * 1024 / 1e6, len(vals)))
PY
  rm -rf $O/${v}_FETCH_SIZE $O/${v}_WRITE_SIZE
done
