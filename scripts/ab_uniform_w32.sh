R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for wl in "17 1000" "26 1000" "31 1000"; do set -- $wl
  for lib in default uw32; do
    if [ $lib = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$lib.so; fi
    python bench.py --workload int64_gt --int-bits $1 --int-base $2 --no-secondary --no-cpu-baseline --steps 5 --warmup 2 --full-line 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('int64 W=$1 %-8s hot %.2f us  cold %.2f us  (hits %s)' % ('$lib', (r.get('kernel_ms_hot') or 0)*1e3, (r.get('kernel_ms_l3_cold') or r['kernel_ms'])*1e3, d['config'].get('hits')))"
  done
done
