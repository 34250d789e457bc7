#!/bin/bash
# round 5, final evidence batch: (1) GPU suite with the measured errors recorded; (2) kernel trace + PMC passes of the level-0 mean-field
# kernels; (3) per-kernel table of one replayed step + of the collate graph; (4) whole-step HBM traffic; (5) the full default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r5k}
part=${2:-all}
mkdir -p gpurun_out/$tag
if [ $part = all ] || [ $part = tests ]; then
CRFCONV_TOL_RECORD=$GRAFT_REPO_ROOT/gpurun_out/$tag/tol_recorded.json timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/$tag/tests.log 2>&1
echo "tests rc=$?"; grep -E "^FAILED|passed|failed" gpurun_out/$tag/tests.log | tail -5 | cut -c1-200
fi
if [ $part = all ] || [ $part = mf ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/mftrace -o t -- python3 scratch/mf_pmc.py > gpurun_out/$tag/mftrace.log 2>&1
find gpurun_out/$tag/mftrace -type f ! -name '*kernel_trace.csv' -delete      # (the trace CSV travels back: scratch/mf_rocprof_json.py runs where profiles/ is tracked)
bash scratch/pmc.sh gpurun_out/$tag/mfpmc 'sim_step_fast|step_fast|bwd_rev|bwd_edge_all' scratch/mf_pmc.py > gpurun_out/$tag/mfpmc_summary.txt 2>&1; echo "mf pmc done"; tail -2 gpurun_out/$tag/mfpmc_summary.txt | cut -c1-160
rm -rf gpurun_out/$tag/mfpmc
fi
if [ $part = all ] || [ $part = tables ]; then
out=gpurun_out/$tag
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $out/trace.log 2>&1 || { tail -20 $out/trace.log | cut -c1-300; }
python3 scratch/step_table.py $out/trace 400 10 > $out/step_table.txt 2>&1; head -3 $out/step_table.txt
rm -rf $out/trace
bash scratch/run_collate_table.sh $tag | head -3
fi
if [ $part = all ] || [ $part = steppmc ]; then
bash scratch/run_step_pmc.sh > gpurun_out/$tag/step_pmc.log 2>&1; echo "step pmc done"; tail -3 gpurun_out/$tag/step_pmc.log | cut -c1-200
rm -rf gpurun_out/step_pmc/FETCH_SIZE_* gpurun_out/step_pmc/WRITE_SIZE_*
fi
if [ $part = all ] || [ $part = pmc2 ]; then
bash scratch/run_pcpmc.sh > gpurun_out/$tag/pcpmc.log 2>&1; echo "pointconv pmc done"; rm -rf gpurun_out/pmc_pc
bash scratch/pmc_mfma.sh gpurun_out/$tag/pmc_mfma > gpurun_out/$tag/pmc_mfma_summary.txt 2>&1; echo "mfma pmc done"; tail -2 gpurun_out/$tag/pmc_mfma_summary.txt | cut -c1-160
rm -rf gpurun_out/$tag/pmc_mfma
fi
if [ $part = all ] || [ $part = bench ]; then
timeout -k 10 600 python3 bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc=$?"
fi
