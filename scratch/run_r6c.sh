#!/bin/bash
cd $GRAFT_REPO_ROOT
tag=${1:-r6c}
mkdir -p gpurun_out/$tag
timeout -k 10 600 python3 bench.py --steps 20 --no-cpu-baseline --no-other-configs > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
r = json.load(open('gpurun_out/$tag/bench.json'))
f = r.get('fresh_batch_replay', r)
print({k: v for k, v in f.items() if 'refresh' in k})
print({k: v for k, v in r['reference_loop'].items() if 'ms_per_step' in k})
PY
