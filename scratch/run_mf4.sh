#!/bin/bash
# forward + backward level-0 timings, mean-field parity tests, kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mf4
out=gpurun_out/mf4
set -o pipefail
timeout -k 10 600 python3 scratch/mf_fwd_time.py 2>&1 | grep -v amdgpu.ids | tee $out/fwd.log || exit 1
H=8 T=3 timeout -k 10 300 python3 scratch/mfb3_bench.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee $out/bwd.log || exit 1
timeout -k 10 900 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "meanfield or crf or golden" 2>&1 | tail -5 | tee $out/pytest.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 scratch/mf_fwd_time.py > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
python3 - <<'PY' | tee -a gpurun_out/mf4/fwd.log
import csv, glob
for f in glob.glob('gpurun_out/mf4/trace/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:40]:
        if 'step' in r['Name'] or 'rev' in r['Name'] or 'edge' in r['Name']:
            print('%-90s calls %6s avg %9.2f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
