#!/bin/bash
# round 6 closing batch: GPU suite with recorded errors + parity report, per-kernel table of one replayed step, whole-step traffic passes,
# the default bench line, one-rank RCCL line, eager host profile, smoke.  usage: scratch/run_r6_final.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r6z}
bash scratch/run_r6_evidence.sh $tag tests
bash scratch/run_r6_evidence.sh $tag tables
bash scratch/run_r6_evidence.sh $tag steppmc
t0=$(date +%s)
bash scratch/run_r6_evidence.sh $tag bench
echo "bench wall $(( $(date +%s) - t0 )) s"
bash scratch/run_r6_evidence.sh $tag spawn
timeout -k 10 300 python3 scratch/eager_profile.py > gpurun_out/$tag/eager_profile.txt 2>&1; grep -E "ms per step|tottime" -A6 gpurun_out/$tag/eager_profile.txt | head -12 | cut -c1-160
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$tag/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/$tag/smoke.txt | cut -c1-200
