#!/bin/bash
# per-kernel times of the classifier-head micro-benchmark (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/head
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/head/prof -o t -- python3 scratch/head_bench.py $1 > gpurun_out/head/prof.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/head/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-70s calls %6s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
rm -rf gpurun_out/head/prof
