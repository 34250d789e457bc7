#!/bin/bash
# round 6 closing batch: the GPU suite (recorded errors, parity report), the default bench line (timed), the eager host profile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r6z}
mkdir -p gpurun_out/$tag
bash scratch/run_r6_evidence.sh $tag tests
t0=$(date +%s); bash scratch/run_r6_evidence.sh $tag bench; echo "bench wall $(( $(date +%s) - t0 )) s"
timeout -k 10 300 python3 scratch/eager_profile.py > gpurun_out/$tag/eager_profile.txt 2>&1; head -3 gpurun_out/$tag/eager_profile.txt
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
