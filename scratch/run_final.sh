#!/bin/bash
# full GPU suite, bench line (with CPU baseline), step table and collate table of the final build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r3f}
out=gpurun_out/$tag
mkdir -p $out
set -o pipefail
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6 | tee $out/pytest.log || exit 1
timeout -k 10 600 python bench.py > $out/bench.log 2>$out/bench.err || { tail -20 $out/bench.err; exit 1; }
grep '^{' $out/bench.log | tail -1 > $out/bench_line.json
python3 -c "
import json
r = json.load(open('$out/bench_line.json'))
print('value %.2f M pts/s  %.3f ms/step  pipelined %.3f ms  collate graph %.3f ms' % (r['value'], r['ms_per_step'], r['pipelined_ms_per_batch'], r['preprocess_plus_refresh_graph_ms_per_batch']))
for k in ('roofline', 'roofline_bwd', 'roofline_fwd_bwd', 'roofline_pointconv', 'cpu_baseline'):
    print(k, json.dumps(r.get(k))[:400])
" | tee $out/summary.log
