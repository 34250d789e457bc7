#!/bin/bash
cd $GRAFT_REPO_ROOT
tag=${1:-r6e}
mkdir -p gpurun_out/$tag
rm -f gpurun_out/$tag/parity_report.json
CRFCONV_PARITY_RECORD=$GRAFT_REPO_ROOT/gpurun_out/$tag/parity_report.json timeout -k 10 900 python -m pytest tests/test_gpu_model.py -q -x -k "config or block_resident or meanfield" > gpurun_out/$tag/tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/$tag/tests.log | tail -5 | cut -c1-250
timeout -k 10 200 python3 scratch/mf_block_ab.py 3 > gpurun_out/$tag/ab_T3.txt 2>&1; tail -6 gpurun_out/$tag/ab_T3.txt
timeout -k 10 200 python3 scratch/mf_block_ab.py 1 > gpurun_out/$tag/ab_T1.txt 2>&1; tail -5 gpurun_out/$tag/ab_T1.txt
timeout -k 10 200 python3 scratch/mf_block_ab.py 5 > gpurun_out/$tag/ab_T5.txt 2>&1; tail -5 gpurun_out/$tag/ab_T5.txt
timeout -k 10 200 python3 scratch/mf_block_stamps.py 3 > gpurun_out/$tag/stamps_T3.txt 2>&1; tail -36 gpurun_out/$tag/stamps_T3.txt
