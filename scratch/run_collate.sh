cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-collate}
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 scratch/collate_run.py > $out/run.log 2>&1 || { tail -20 $out/run.log; exit 1; }
python3 scratch/collate_table.py $out/trace > $out/collate_table.txt 2>&1; head -70 $out/collate_table.txt
