# Instructions per phase of k_like_lean: SQ counters of the variant builds that leave the kernel early (LC_LEAN_STOP)
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in stopm2 stop1 stop2 stop3 default; do
  if [ "$v" = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$v.so; fi
  GROUPS_SEL="1 2" bash $R/scripts/pmc_sq.sh lean_$v --no-needle-classes 2>/dev/null | grep "k_like_lean\|like_lean" | awk -v v=$v '{print v, $0}'
done
