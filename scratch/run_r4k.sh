#!/bin/bash
# round 4, final evidence batch: GPU suite with the measured errors recorded, MFMA counters of the step (incl. the classifier-head
# kernels), whole-step HBM traffic, the full bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
CRFCONV_TOL_RECORD=$GRAFT_REPO_ROOT/gpurun_out/r4k/tol_recorded.json timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4k/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r4k/tests.log
bash scratch/pmc_mfma.sh gpurun_out/pmc_mfma5 > gpurun_out/pmc_mfma5_summary.txt 2>&1; echo "mfma pmc done"; tail -3 gpurun_out/pmc_mfma5_summary.txt
rm -rf gpurun_out/pmc_mfma5
bash scratch/run_step_pmc.sh > gpurun_out/r4k/step_pmc.log 2>&1; echo "step pmc done"; tail -3 gpurun_out/r4k/step_pmc.log
rm -rf gpurun_out/step_pmc/FETCH_SIZE_* gpurun_out/step_pmc/WRITE_SIZE_*
timeout -k 10 600 python3 bench.py > gpurun_out/r4k/bench.json 2> gpurun_out/r4k/bench.err; echo "bench rc=$?"
