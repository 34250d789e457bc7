#!/bin/bash
# round 6 work batch b: new pipeline / vote / dilation tests, the load_ host profile, one full bench line
cd $GRAFT_REPO_ROOT
tag=${1:-r6b}
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -m pytest tests/test_gpu_eval.py tests/test_gpu_sparse.py tests/test_gpu_dist.py tests/test_gpu_model.py -q -x -k "tiled_scene or votes_merged or scene_crops or dilated or graph_builders or static_batch or tables_built or deferred_table or table_rejects" > gpurun_out/$tag/tests.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/$tag/tests.log | cut -c1-250
timeout -k 10 300 python3 scratch/load_profile.py > gpurun_out/$tag/load_profile.txt 2>&1; head -4 gpurun_out/$tag/load_profile.txt; sed -n '/cumulative/,$p' gpurun_out/$tag/load_profile.txt | head -34 | cut -c1-150
CRFCONV_BENCH_PROFILE_LOAD=1 timeout -k 10 900 python3 bench.py --steps 30 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc=$?"; grep -A32 "cumulative" gpurun_out/$tag/bench.err | cut -c1-300
python3 - <<PY
import json
r = json.load(open('gpurun_out/$tag/bench.json'))
print('ms_per_step', r['ms_per_step'], 'value', r['value'])
print('roofline', {k: r['roofline'].get(k) for k in ('form', 'avg_launch_us', 'frac', 'frac_rocprof', 'other_form_launch_us', 'traffic')})
rl = r.get('reference_loop') or {}
print('reference_loop', {k: v for k, v in rl.items() if 'ms_per_step' in k or k == 'error'})
oc = r.get('other_configs') or {}
print('pipelines', json.dumps(oc.get('pipelines', oc.get('error')), indent=1)[:2500])
print('table_refresh', r.get('table_refresh_ms_per_batch'), 'pipelined', r.get('pipelined_ms_per_batch'))
PY
