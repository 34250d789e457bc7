#!/bin/bash
cd $GRAFT_REPO_ROOT
tag=${1:-r6i}
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -m pytest tests/test_gpu_eval.py tests/test_gpu_dist.py -q -x -k "tiled_scene or scene_crops" > gpurun_out/$tag/tests.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/$tag/tests.log | cut -c1-250
timeout -k 10 600 python3 - > gpurun_out/$tag/pipelines.json 2> gpurun_out/$tag/pipelines.err <<'PY'
import json, sys, time
sys.path.insert(0, '.')
import crfconv_amd, torch
from benchlib.configs import config_pipelines
t0 = time.time()
out = config_pipelines(torch.device('cuda', 0))
out['wall_s'] = time.time() - t0
print(json.dumps(out, indent=1))
PY
echo "pipelines rc=$?"; tail -3 gpurun_out/$tag/pipelines.err | cut -c1-300; python3 -c "
import json; r=json.load(open('gpurun_out/$tag/pipelines.json')); c=r['C5 Semantic3D-like scene, tiled inference']; print({k: c[k] for k in c if 'ms' in k or 'points_per_s' in k}); print(r['wall_s'])"
