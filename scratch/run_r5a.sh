#!/bin/bash
# round 5 work batch: GPU suite (stop at the first failure), one short bench line, the per-kernel table + launch sequence of one replayed step
cd $GRAFT_REPO_ROOT
tag=${1:-r5a}
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/$tag/tests.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|passed|failed" gpurun_out/$tag/tests.log | tail -8 | cut -c1-200
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 40 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { echo "bench failed"; tail -5 gpurun_out/$tag/bench.err | cut -c1-300; }
python3 -c "
import json; r=json.load(open('gpurun_out/$tag/bench.json')); print('ms_per_step %.4f  value %.2f  pipelined %.3f  eager %.2f captured %.2f' % (r['ms_per_step'], r['value'], r['pipelined_ms_per_batch'] or 0, r['trainval_eager_ms_per_step'] or 0, r['trainval_captured_ms_per_step'] or 0))"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $out/trace.log 2>&1 || { tail -20 $out/trace.log | cut -c1-300; exit 1; }
python3 scratch/step_table.py $out/trace 400 10 > $out/step_table.txt 2>&1; head -3 $out/step_table.txt
rm -rf $out/trace
# optional A/B behind the batch: the variants listed in scratch/ab_next.txt (one line, space separated)
if [ -s scratch/ab_next.txt ]; then
  bash scratch/run_ab.sh $(cat scratch/ab_next.txt) > gpurun_out/$tag/ab.log 2>&1; tail -12 gpurun_out/$tag/ab.log | cut -c1-160
fi
