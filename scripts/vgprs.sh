#!/bin/bash
# usage: scripts/vgprs.sh <file.hip> <symbol substring> [extra hipcc flags]: the functions / kernels with the most VGPRs and
# every one that uses scratch memory (from the resource symbols of the assembly; name holds the substring).
R=$(cd "$(dirname "$0")/.." && pwd)
f=$1; pat=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S "$@" -o /tmp/vgprs.asm $R/liquid_cache_amd/csrc/$f 2>/dev/null || { echo compile failed; exit 1; }
short() { sed -E 's/^\s*\.set (\.L)?//; s/_ZN2lc12_GLOBAL__N_1[0-9]+//; s/EEEjNS0_12RegEntryArgsE//'; }
echo "VGPRs (top 12):"; grep -E "^\s*\.set [^,]*$pat[^,]*\.num_vgpr, [0-9]+$" /tmp/vgprs.asm | short | sed 's/\.num_vgpr, / /' | sort -k2 -n | tail -12 | tr '\n' ';'; echo
echo "scratch:"; grep -E "^\s*\.set [^,]*$pat[^,]*\.private_seg_size, [1-9][0-9]*$" /tmp/vgprs.asm | short | sed 's/\.private_seg_size, / /' | tr '\n' ';'; echo
