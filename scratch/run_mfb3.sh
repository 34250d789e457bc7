#!/bin/bash
# one GPU session: parity of the new mean-field backward on hostile shapes, timing of the default build and of the A/B builds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mfb3
out=gpurun_out/mfb3
set -o pipefail
echo "== parity (hub rows, small)" | tee $out/parity.log
env H=8 T=3 B=2 N=3000 NOTIME=1 timeout -k 10 500 python3 scratch/mfb3_bench.py 2>&1 | tail -4 | tee -a $out/parity.log || exit 1
for cfg in "H=8 T=1" "H=16 T=3" "H=32 T=2" "H=8 T=5 KNN=32"; do
  echo "-- $cfg HUB" | tee -a $out/parity.log
  env $cfg B=2 N=2048 HUB=1 NOTIME=1 timeout -k 10 120 python3 scratch/mfb3_bench.py 2>&1 | tail -8 | tee -a $out/parity.log || exit 1
done
echo "== timing, default build" | tee $out/time.log
H=8 T=3 timeout -k 10 200 python3 scratch/mfb3_bench.py 2>&1 | tail -9 | tee -a $out/time.log || exit 1
H=16 T=3 N=10240 timeout -k 10 200 python3 scratch/mfb3_bench.py 2>&1 | tail -5 | tee -a $out/time.log || exit 1
for v in c4 c4t2 c2t2 f8 f8t1 f4t1 enb8; do
  echo "== variant $v" | tee -a $out/time.log
  CRFCONV_LIB=$PWD/scratch/variants/lib_$v.so H=8 T=3 timeout -k 10 200 python3 scratch/mfb3_bench.py 2>&1 | tail -5 | tee -a $out/time.log || exit 1
done
echo "== kernel trace" | tee -a $out/time.log
H=8 T=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 scratch/mfb3_bench.py > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
python3 - <<'PY' | tee -a gpurun_out/mfb3/time.log
import csv, glob
for f in glob.glob('gpurun_out/mfb3/trace/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print('%-90s calls %6s avg %9.2f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
