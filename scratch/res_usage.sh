#!/bin/bash
# usage: scratch/res_usage.sh <file.hip> [extra flags]  -- registers / scratch / LDS of every kernel of one translation unit
cd /root/repo/crfconv_amd/csrc
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/res_usage.o 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|VGPRs Spill|LDS Size|error|warning:" | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' | paste - - - - - | sed -e 's/Function Name: //' | c++filt | cut -c1-260
