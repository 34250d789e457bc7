#!/bin/bash
# PMC passes + kernel trace of the level-0 mean-field forward + backward (final build of the round)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_r4a
bash scratch/pmc.sh gpurun_out/pmc_r4a 'sim_step_fast|step_fast|bwd_rev|bwd_edge_all' scratch/mf_pmc.py > gpurun_out/pmc_r4a_summary.txt 2>&1
tail -5 gpurun_out/pmc_r4a_summary.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_r4a/trace -o t -- python3 scratch/mf_pmc.py > gpurun_out/pmc_r4a/trace.log 2>&1
python3 - <<'PY' > gpurun_out/pmc_r4a_kernel_stats.txt
import csv, glob
for f in glob.glob('gpurun_out/pmc_r4a/trace/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:40]:
        if r['Name'].startswith('void crf::') or r['Name'].startswith('crf::'):
            print('| %s | %s | %.2f |' % (r['Name'].split('(')[0].replace('void ', ''), r['Calls'], float(r['AverageNs']) / 1e3))
PY
cat gpurun_out/pmc_r4a_kernel_stats.txt
