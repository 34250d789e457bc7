#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5e/tests.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|passed|failed" gpurun_out/r5e/tests.log | tail -8 | cut -c1-200
bash scratch/run_ab.sh wideoff pf4 pf4wideoff > gpurun_out/r5e/ab.log 2>&1; tail -8 gpurun_out/r5e/ab.log | cut -c1-200
