#!/bin/bash
# usage: scratch/run_ab.sh <variant> [<variant> ...]   -- ms_per_step of the default library and of scratch/variants/lib_<variant>.so, same box, twice each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in default "$@"; do
    if [ "$v" = default ]; then unset CRFCONV_LIB; else export CRFCONV_LIB=$GRAFT_REPO_ROOT/scratch/variants/lib_$v.so; fi
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 40 > gpurun_out/ab/$v.$rep.json 2> gpurun_out/ab/$v.$rep.err || { echo "$v failed"; tail -5 gpurun_out/ab/$v.$rep.err; }
    python3 -c "
import json,sys; r=json.load(open('gpurun_out/ab/$v.$rep.json')); print('%-12s rep $rep  ms_per_step %.4f  value %.2f' % ('$v', r['ms_per_step'], r['value']))"
  done
done
