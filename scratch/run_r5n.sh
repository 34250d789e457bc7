#!/bin/bash
# round 5 work batch: the graphed-module test, the join A/B, one bench line with the reference-loop numbers
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "graphed or captured_step or mlp or join or resnet" > gpurun_out/r5n/tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5n/tests.log | cut -c1-300
bash scratch/run_ab.sh arg:--no-fold-join bt24 > gpurun_out/r5n/ab.log 2>&1; tail -8 gpurun_out/r5n/ab.log | cut -c1-160
python3 -c "
import json; r=json.load(open('gpurun_out/ab/default.2.json')); print({k: r[k] for k in r if k.startswith('trainval')})"
