#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/srd
timeout -k 10 300 python3 scratch/mf_levels.py --bwd 2>&1 | grep -v amdgpu.ids | tee gpurun_out/srd/default.log || exit 1
CRFCONV_LIB=$GRAFT_REPO_ROOT/scratch/variants/lib_nosrd.so timeout -k 10 300 python3 scratch/mf_levels.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/srd/nosrd.log || exit 1
timeout -k 10 300 python3 scratch/mf_levels.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/srd/default2.log || exit 1
timeout -k 10 900 python3 -X faulthandler -m pytest tests/test_gpu_model.py -x -q -m gpu -k "captured_step or two_consumers or skips_the_update or collate_graph_replay or meanfield" > gpurun_out/srd/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/srd/tests.log
