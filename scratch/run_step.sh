#!/bin/bash
# model parity tests, bench line, and the per-kernel table of one replayed training step (full, uncut)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r3}
mkdir -p gpurun_out/$tag
out=gpurun_out/$tag
set -o pipefail
timeout -k 10 1000 python -m pytest tests/test_gpu_model.py tests/test_gpu_sparse.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6 | tee $out/pytest.log || exit 1
timeout -k 10 400 python bench.py --no-cpu-baseline > $out/bench.log 2>$out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json
r = json.loads([l for l in open('$out/bench.log') if l.startswith('{')][0])
print('value %.2f M pts/s  %.3f ms/step  pipelined %.3f ms  fwd %.1f us  bwd %.1f us  pc %s' % (r['value'], r['ms_per_step'], r['pipelined_ms_per_batch'], r['roofline']['avg_launch_us'], r['roofline_bwd']['avg_launch_us'], r.get('roofline_pointconv')))
" | tee $out/summary.log
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
python3 scratch/step_table.py $out/trace 400 10 > $out/step_table.txt 2>&1; head -12 $out/step_table.txt
