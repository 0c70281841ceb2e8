# A/B of the many-candidate LIKE walkers: variants built with `make VARIANT=... EXTRA=...` (liquid_cache_amd/variants/)
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in default "$@"; do
  if [ "$v" = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$v.so; fi
  python $R/bench.py --full-line --secondary-set like --no-cpu-baseline --no-needle-classes --steps 5 --warmup 2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for k,v in d['secondary'].items():
    print('$v', k, 'cold %.1f hot %.1f us'%(v['kernel_ms']*1e3, v['kernel_ms_hot']*1e3), v.get('hits'))
"
done
