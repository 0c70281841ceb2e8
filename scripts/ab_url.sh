#!/bin/bash
# A/B of headline-kernel variants on one box: scripts/ab_url.sh <out-dir> <variant.so|default> ...
out=$1; shift
mkdir -p "$out"
for v in "$@"; do
    name=$(basename "$v" .so)
    if [ "$v" = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH="$PWD/$v"; fi
    timeout 200 python bench.py --full-line --no-secondary --no-cpu-baseline > "$out/url_$name.json" 2> "$out/url_$name.err"
    echo "== $name"; python scripts/bench_summary.py "$out/url_$name.json" 2>&1 | head -2
done
unset LC_LIB_PATH
