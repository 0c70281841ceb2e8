# many-candidate LIKE (no signature index): kernel time under ablation flags of the timing build
R=${GRAFT_REPO_ROOT:-/root/repo}
export LC_LIB_PATH=$R/liquid_cache_amd/libliquid_cache_amd_timing.so
for f in ${FLAGS:-0 1 64 128 256 384}; do
  LC_DEBUG_FLAGS=$f python $R/bench.py --full-line --no-signatures --no-secondary --no-cpu-baseline --no-cold --no-needle-classes --steps 5 --warmup 2 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags $f kernel_ms', d['roofline']['kernel_ms'], d['config']['hits'])"
done
