#!/bin/bash
# round 6: a test selection, then the A/B of the training step against environment switches.  usage: run_r6_one.sh <tag> "<pytest -k expr>" <variant> ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; sel=$2; shift 2
mkdir -p gpurun_out/$tag
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "$sel" > gpurun_out/$tag/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/$tag/tests.log | cut -c1-300
[ $rc = 0 ] || exit $rc
bash scratch/run_ab.sh "$@" 2>&1 | tee gpurun_out/$tag/ab.txt
