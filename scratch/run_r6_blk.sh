#!/bin/bash
# round 6: the block-resident mean-field forward -- its tests, then the A/B against the per-step launches
cd $GRAFT_REPO_ROOT
tag=${1:-r6blk}
mkdir -p gpurun_out/$tag
timeout -k 10 500 python -m pytest tests/test_gpu_model.py -q -x -k "block_resident or riders_in_a_statistics or meanfield_vs_oracle or crf_matrices_riding" > gpurun_out/$tag/tests.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/$tag/tests.log | cut -c1-250
timeout -k 10 200 python3 scratch/mf_block_ab.py 3 > gpurun_out/$tag/ab_T3.txt 2>&1; cat gpurun_out/$tag/ab_T3.txt | tail -8
timeout -k 10 200 python3 scratch/mf_block_ab.py 1 > gpurun_out/$tag/ab_T1.txt 2>&1; tail -5 gpurun_out/$tag/ab_T1.txt
timeout -k 10 200 python3 scratch/mf_block_ab.py 5 > gpurun_out/$tag/ab_T5.txt 2>&1; tail -5 gpurun_out/$tag/ab_T5.txt
timeout -k 10 200 python3 scratch/mf_block_ab.py 3 none > gpurun_out/$tag/ab_T3_unsorted.txt 2>&1; tail -5 gpurun_out/$tag/ab_T3_unsorted.txt
timeout -k 10 200 python3 scratch/mf_block_stamps.py 3 > gpurun_out/$tag/stamps_T3.txt 2>&1; cat gpurun_out/$tag/stamps_T3.txt | tail -40
