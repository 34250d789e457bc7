#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5c/tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5c/tests.log | cut -c1-300
bash scratch/run_ab.sh arg:--no-side-jobs rows64 arg:--side-us=3 arg:--mfma-min-rows=8192 > gpurun_out/r5c/ab.log 2>&1; tail -12 gpurun_out/r5c/ab.log | cut -c1-200
