#!/bin/bash
# grids of the classifier-head kernels (csrc/head.hip), one box: scratch/head_bench.py per setting
cd $GRAFT_REPO_ROOT
for v in "" "CRFCONV_HD_GRID=768" "CRFCONV_HD_GRID=1024" "CRFCONV_HD_GRID_ST=384" "CRFCONV_HD_GRID_ST=512" "CRFCONV_HD_GRID_DX=768" "CRFCONV_HD_GRID_DX=1024" "CRFCONV_HD_GRID_P1=128" "CRFCONV_HD_GRID_P1=512"; do
  echo "== ${v:-default}"
  env $v timeout -k 10 120 python3 scratch/head_bench.py recompute 2>&1 | grep -v amdgpu.ids
done
