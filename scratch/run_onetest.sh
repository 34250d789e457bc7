#!/bin/bash
# one test selection under the default library and under variants, with the measured errors printed.  usage: run_onetest.sh "<-k expr>" <variant> ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sel=$1; shift
for v in default "$@"; do
  unset CRFCONV_LIB
  [ $v = default ] || export CRFCONV_LIB=$GRAFT_REPO_ROOT/scratch/variants/lib_$v.so
  echo "== $v"
  CRFCONV_TEST_REPORT=1 CRFCONV_TOL_RECORD=/tmp/tolrec.json timeout -k 10 300 python3 -m pytest tests -m gpu -q -x -s -k "$sel" > /tmp/onetest.log 2>&1
  grep -E "passed|failed" /tmp/onetest.log | tail -1
  grep "assert_close" /tmp/onetest.log | awk '{for(i=1;i<=NF;i++) if($i=="err") print $(i+1), $0}' | sort -g | tail -${TOPN:-6} | cut -c1-200
done
