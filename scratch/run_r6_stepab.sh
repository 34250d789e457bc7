#!/bin/bash
# round 6: the training step with the block-resident mean-field forward on / off (two interleaved runs each)
cd $GRAFT_REPO_ROOT
tag=${1:-r6stepab}
mkdir -p gpurun_out/$tag
for i in 1 2; do
for mode in off auto; do
CRFCONV_MF_BLOCK=$mode timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 40 > gpurun_out/$tag/bench_$mode$i.json 2> gpurun_out/$tag/bench_$mode$i.err || { echo "bench $mode failed"; tail -5 gpurun_out/$tag/bench_$mode$i.err | cut -c1-300; }
python3 -c "
import json; r=json.load(open('gpurun_out/$tag/bench_$mode$i.json')); print('$mode: ms_per_step %.4f  value %.2f  mf fwd %.2f us (frac %.3f)  layer fwd %s' % (r['ms_per_step'], r['value'], r['roofline']['avg_launch_us'], r['roofline']['frac'], r.get('roofline_layer', {}).get('fwd_us')))"
done
done
