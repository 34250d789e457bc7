#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout -k 10 900 python3 -X faulthandler -m pytest tests/test_gpu_model.py tests/test_gpu_sparse.py -x -q -m gpu -k "pointconv or deferred or replayed or golden or resblock or config or sparse or depthwise" > gpurun_out/r4g/tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r4g/tests.log
bash scratch/run_ab.sh env:CRFCONV_NO_WIDE_MFMA=1
