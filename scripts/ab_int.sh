# A/B of the fixed-width predicate kernels: variants built with `make VARIANT=... EXTRA=...`
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in default "$@"; do
  if [ "$v" = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$v.so; fi
  python $R/bench.py --full-line --workload int64_gt --int-kind date32 --int-bits 12 --int-base 8036 --secondary-set int --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']; print('$v primary date32 cold %.1f hot %.1f step %.1f'%(r['kernel_ms']*1e3, r['kernel_ms_hot']*1e3, d['ms_per_step']*1e3))
for k,v in d['secondary'].items():
    if 'kernel_ms' in v: print('$v', k, 'cold %.1f hot %.1f us'%(v['kernel_ms']*1e3, v['kernel_ms_hot']*1e3), v.get('hits'))
"
done
