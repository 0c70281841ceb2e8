# A/B of the Q6 chain (BASELINE config 4 at 600 M rows) over variant builds
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in default "$@"; do
  if [ "$v" = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$v.so; fi
  python $R/bench.py --full-line --workload tpch_q6 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v q6 step %.1f us kernel %.1f us frac %.3f count %s'%(d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['frac'], d['config'].get('hits')))"
done
