# get-with-selection and the q21 pipeline (bench secondaries) for the default build and variant builds
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in default "$@"; do
  if [ "$v" = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$v.so; fi
  python $R/bench.py --full-line --secondary-set q21,int --no-cpu-baseline --no-needle-classes --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); g=d['secondary']['int64_gt_w62']['get_with_selection']
print('$v 10pct ms %.4f frac %.3f | 0.1pct ms %.4f'%(g['10pct']['ms'], g['10pct']['frac'], g['0.1pct']['ms']))
q=d['secondary']['q21_pipeline']; print('$v q21 ms %.4f with partials %.4f'%(q['ms'], q['ms_with_group_partials']))"
done
