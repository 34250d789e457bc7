#!/bin/bash
# usage: scratch/res_usage.sh <file.hip> [filter] [-DFLAG ...]  -- VGPRs / scratch / LDS / occupancy of every kernel of one translation unit
f=$1; shift; flt=${1:-.}; shift
cd "$(dirname "$0")/../crfconv_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed "$@" -c $f -o /tmp/res_usage.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - - \
 | sed -E 's/Function Name: //' | while IFS=$'\t' read n a b c d e; do echo "$(echo $n | c++filt | sed -E 's/\(.*//; s/crf:://' | cut -c1-70) | $a | $b | $c | $d | $e"; done | grep -E "$flt"
