#!/bin/bash
# tests of the new pieces first (fast failures), then the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout -k 10 900 python3 -X faulthandler -m pytest tests/test_gpu_model.py -x -q -m gpu -k "captured_step or two_consumers or skips_the_update or collate_graph_replay" > gpurun_out/r4a/new_tests.log 2>&1
echo "new tests rc=$?"; tail -15 gpurun_out/r4a/new_tests.log
timeout -k 10 600 python3 -X faulthandler bench.py --no-cpu-baseline > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
echo "bench rc=$?"; tail -c 1500 gpurun_out/r4a/bench.err
