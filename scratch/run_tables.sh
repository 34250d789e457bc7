cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r3f}
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
python3 scratch/step_table.py $out/trace 400 10 > $out/step_table.txt 2>&1; head -5 $out/step_table.txt
rm -rf $out/trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/ctrace -o t -- python3 scratch/collate_run.py > $out/crun.log 2>&1 || { tail -20 $out/crun.log; exit 1; }
python3 scratch/collate_table.py $out/ctrace > $out/collate_table.txt 2>&1; head -5 $out/collate_table.txt
rm -rf $out/ctrace
