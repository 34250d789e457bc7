#!/bin/bash
# "last workgroup finishes" sums (PointConv statistics / coefficients, MLP backward channel part): tests, then the step with and without (one box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/ticket_tests.log 2>&1 || { tail -30 gpurun_out/ticket_tests.log; exit 1; }
tail -2 gpurun_out/ticket_tests.log
bash scratch/run_ab.sh env:CRFCONV_NO_PC_TICKET=1 env:CRFCONV_NO_MLP_TICKET=1
