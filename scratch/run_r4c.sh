#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout -k 10 900 python3 -m pytest tests/test_gpu_native.py tests/test_gpu_eval.py -x -q -m gpu > gpurun_out/r4c/native_tests.log 2>&1
echo "native/eval tests rc=$?"; tail -4 gpurun_out/r4c/native_tests.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "crf or meanfield or golden or replayed" > gpurun_out/r4c/model_tests.log 2>&1
echo "model tests rc=$?"; tail -4 gpurun_out/r4c/model_tests.log
timeout -k 10 300 python3 scratch/mf_levels.py --bwd 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c/default.log || exit 1
CRFCONV_LIB=$GRAFT_REPO_ROOT/scratch/variants/lib_nosrd_bwd.so timeout -k 10 300 python3 scratch/mf_levels.py --bwd 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c/nosrd_bwd.log || exit 1
timeout -k 10 600 python3 bench.py > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err
echo "bench rc=$?"; python3 -c "
import json; r=json.load(open('gpurun_out/r4c/bench.json'))
print({k: r[k] for k in ('value','ms_per_step','pipelined_ms_per_batch','trainval_eager_ms_per_step','trainval_captured_ms_per_step')})
print(r['cpu_baseline']['grid_subsample']['sample']); print(r['cpu_baseline']['knn']['sample'])"
