#!/bin/bash
# usage: scratch/run_ab.sh <variant> ...   -- ms_per_step of the default build against variants on ONE box, twice each, interleaved.
#   <name>          the library scratch/variants/lib_<name>.so (scratch/build_variant.sh)
#   env:VAR=VALUE   the default library with one environment switch
#   arg:--flag=V    the default library with one more bench.py argument
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in default "$@"; do
    tag=$(echo "$v" | tr ':=' '__')
    unset CRFCONV_LIB
    pre=""
    extra=""
    case "$v" in
      default) ;;
      env:*) pre="${v#env:}" ;;
      arg:*) extra="${v#arg:}" ;;
      *) export CRFCONV_LIB=$GRAFT_REPO_ROOT/scratch/variants/lib_$v.so ;;
    esac
    env $pre timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 40 $extra > gpurun_out/ab/$tag.$rep.json 2> gpurun_out/ab/$tag.$rep.err || { echo "$v failed"; tail -3 gpurun_out/ab/$tag.$rep.err | cut -c1-300; continue; }
    python3 -c "
import json,sys; r=json.load(open('gpurun_out/ab/$tag.$rep.json')); print('%-36s rep $rep  ms_per_step %.4f  value %.2f  pipelined %.3f' % ('$v', r['ms_per_step'], r['value'], r['pipelined_ms_per_batch'] or 0))"
  done
done
