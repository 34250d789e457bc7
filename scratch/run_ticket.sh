#!/bin/bash
# full GPU tests, then the step with and without one switch each (one box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/ticket_tests.log 2>&1 || { tail -30 gpurun_out/ticket_tests.log; exit 1; }
tail -2 gpurun_out/ticket_tests.log
bash scratch/run_ab.sh "$@"
