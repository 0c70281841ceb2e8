#!/bin/bash
# The library's HOST code under AddressSanitizer (device code is not instrumented: -fno-gpu-sanitize).
#   make -C liquid_cache_amd/csrc VARIANT=asan EXTRA="-fsanitize=address -fno-gpu-sanitize -Xarch_host -fno-omit-frame-pointer -Xarch_host -g"
# (frame pointers for the HOST only: with -fno-omit-frame-pointer on the device side the register-resident integer kernels miscompile
# — constant-outcome entries of ALP float columns evaluate a range instead; seen with this round's and with round 5's source alike)
# usage: scripts/asan_suite.sh [pytest args]     default: the CPU suite (-m "not gpu"); on a GPU box pass e.g. tests/test_gpu_round6.py -m gpu
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export LC_LIB_PATH=$R/liquid_cache_amd/variants/libliquid_cache_amd_asan.so
[ -e "$LC_LIB_PATH" ] || { echo "build the asan variant first (see the header of this script)"; exit 2; }
if [ $# -eq 0 ]; then set -- tests -m "not gpu"; fi
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0 python -m pytest -q "$@"
