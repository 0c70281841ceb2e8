#!/bin/bash
# A/B of fixed-width kernel variants on one box: scripts/ab_fixed.sh <out-dir> <variant.so|default> ...
# Every variant runs the Date32 W=12 predicate as the main workload plus the integer-column and TPC-H Q6 secondaries.
out=$1; shift
mkdir -p "$out"
for v in "$@"; do
    name=$(basename "$v" .so)
    if [ "$v" = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH="$PWD/$v"; fi
    timeout 300 python bench.py --full-line --workload int64_gt --int-kind date32 --int-bits 12 --no-cpu-baseline \
        --secondary-set int,q6 > "$out/$name.json" 2> "$out/$name.err"
done
unset LC_LIB_PATH
python scripts/ab_summary.py "$out"
