#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r5col}
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/ctrace -o t -- python3 scratch/collate_run.py > $out/crun.log 2>&1 || { tail -20 $out/crun.log | cut -c1-300; exit 1; }
python3 scratch/collate_table.py $out/ctrace > $out/collate_table.txt 2>&1; head -40 $out/collate_table.txt | cut -c1-140
rm -rf $out/ctrace
