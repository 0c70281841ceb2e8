R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for lib in default mb5 mb6; do
  if [ $lib = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$lib.so; fi
  python bench.py --no-signatures --no-secondary --no-cpu-baseline --steps 3 --warmup 1 --scans-per-step 8 --rotate 1 --full-line 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('no_signatures %-8s kernel %s hot %.1f us  cold %.1f us  hits %s' % ('$lib', r['kernel'], (r.get('kernel_ms_hot') or 0)*1e3, (r.get('kernel_ms_l3_cold') or r['kernel_ms'])*1e3, d['config'].get('hits')))"
done
