# A/B of the LIKE needle classes (bench.py --full-line like_needle_classes: masks compared with the CPU oracle) over variant builds
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in default "$@"; do
  if [ "$v" = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$v.so; fi
  python $R/bench.py --full-line --secondary-set like --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v step us %.1f masks ok %s'%(d['ms_per_step']*1e3, d['config'].get('needle_masks_all_match_cpu_oracle')))
for k,v in d['like_needle_classes'].items():
    print('$v', k, v['predicate'], 'hot %.1f cold %.1f'%(v['kernel_us_hot'], v['kernel_us_l3_cold']), v['mask_equals_cpu_oracle'])
for k,v in d['secondary'].items():
    print('$v', k, 'cold %.1f hot %.1f us'%(v['kernel_ms']*1e3, v['kernel_ms_hot']*1e3))
"
done
