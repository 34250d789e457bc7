#!/bin/bash
# round 4 evidence batch: GPU suite with the measured errors recorded, PointConv / MFMA / mean-field counter passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
CRFCONV_TOL_RECORD=$GRAFT_REPO_ROOT/gpurun_out/r4d/tol_recorded.json timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4d/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r4d/tests.log
bash scratch/run_pcpmc.sh > gpurun_out/r4d/pcpmc.log 2>&1; echo "pc pmc done"; tail -3 gpurun_out/r4d/pcpmc.log
bash scratch/pmc_mfma.sh gpurun_out/pmc_mfma4 > gpurun_out/pmc_mfma4_summary.txt 2>&1; echo "mfma pmc done"; tail -3 gpurun_out/pmc_mfma4_summary.txt
bash scratch/run_pmc4.sh > gpurun_out/r4d/mfpmc.log 2>&1; echo "mf pmc done"; tail -3 gpurun_out/r4d/mfpmc.log
