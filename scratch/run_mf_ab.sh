#!/bin/bash
# A/B of the isolated level-0 mean-field kernels (scratch/mf_pmc.py) under rocprofv3 --kernel-trace: the default library, then variants
# (scratch/variants/lib_<name>.so), twice each, interleaved.  usage: scratch/run_mf_ab.sh <tag> "<pytest -k expr | ->" <variant> ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; sel=$2; shift 2
mkdir -p gpurun_out/$tag
if [ "$sel" != "-" ]; then
  timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "$sel" > gpurun_out/$tag/tests.log 2>&1
  rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/$tag/tests.log | cut -c1-300
  [ $rc = 0 ] || exit $rc
fi
for rep in 1 2; do
  for v in default "$@"; do
    unset CRFCONV_LIB
    [ $v = default ] || export CRFCONV_LIB=$GRAFT_REPO_ROOT/scratch/variants/lib_$v.so
    d=gpurun_out/$tag/tr_${v}_$rep
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 scratch/mf_pmc.py 20 ${MF_LEVEL:-0} > $d.log 2>&1 || { echo "$v failed"; tail -5 $d.log | cut -c1-300; exit 1; }
    echo "== $v rep $rep"; python3 scratch/trace_avg.py $d 'bwd_rev|bwd_edge|mf_block|step_fast' | tee -a gpurun_out/$tag/ab.txt
    rm -rf $d
  done
done
