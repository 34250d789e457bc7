#!/bin/bash
# round 4: per-kernel table of one replayed step + collate graph, whole-step HBM traffic, bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scratch/run_tables.sh r4f > gpurun_out/r4f_tables.log 2>&1; echo "tables done"; tail -3 gpurun_out/r4f_tables.log
bash scratch/run_step_pmc.sh > gpurun_out/r4f_step_pmc.log 2>&1; echo "step pmc done"; tail -3 gpurun_out/r4f_step_pmc.log
timeout -k 10 600 python3 bench.py > gpurun_out/r4f_bench.json 2> gpurun_out/r4f_bench.err; echo "bench rc=$?"
