cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r3final2}
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/ctrace -o t -- python3 scratch/collate_run.py > $out/crun.log 2>&1 || { tail -20 $out/crun.log; exit 1; }
python3 scratch/collate_table.py $out/ctrace > $out/collate_table.txt 2>&1; head -8 $out/collate_table.txt
rm -rf $out/ctrace
