#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in ${VARIANTS:-sm1 s1 s2 s3 default}; do
  if [ "$v" = default ]; then unset LC_LIB_PATH; else export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_$v.so; fi
  timeout 120 python scripts/time_like.py "$@" 2>&1 | tail -${TAIL:-1}
done
