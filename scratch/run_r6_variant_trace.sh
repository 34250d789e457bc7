#!/bin/bash
# round 6: A/B of library variants on the training step, then the per-kernel table of one replayed step under the LAST variant.
# usage: run_r6_variant_trace.sh <tag> <grep pattern for the table> <variant> ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; pat=$2; shift 2
mkdir -p gpurun_out/$tag
rm -rf gpurun_out/ab
bash scratch/run_ab.sh "$@" 2>&1 | tee gpurun_out/$tag/ab.txt
for v in default "${@: -1}"; do
  unset CRFCONV_LIB
  [ $v = default ] || export CRFCONV_LIB=$GRAFT_REPO_ROOT/scratch/variants/lib_$v.so
  out=gpurun_out/$tag/$v
  mkdir -p $out
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $out/trace.log 2>&1 || { tail -5 $out/trace.log | cut -c1-300; }
  python3 scratch/step_table.py $out/trace 400 10 > $out/step_table.txt 2>&1
  echo "== $v"; head -1 $out/step_table.txt; grep -E "$pat" $out/step_table.txt
  rm -rf $out/trace
done
