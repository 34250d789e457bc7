#!/bin/bash
# round 6: vector-memory path counters (TA / TCP / TCC) of the level-0 mean-field kernels, one counter set per pass (never with a trace),
# plus a kernel trace of the same script.  usage: scratch/run_r6_ta.sh <tag> [script args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r6ta}; shift
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 -L > $out/counters_all.txt 2>&1
grep -o -E "\b(TA|TCP|TD|TCC|SQ|GRBM)_[A-Za-z0-9_]+" $out/counters_all.txt | sort -u > $out/counter_names.txt; wc -l $out/counter_names.txt
rx='sim_step_fast|step_fast|bwd_rev|bwd_edge_all|mf_block'
i=0
while read -r set; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-include-regex "$rx" --output-format csv -d $out/p$i -- python3 scratch/mf_pmc.py "$@" > $out/p$i.log 2>&1 || echo "pass $i ($set) failed"
done <<'SETS'
TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum
TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR
GRBM_GUI_ACTIVE GRBM_COUNT
SETS
python3 - "$out" <<'PY' > $out/summary.txt
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/*/*counter_collection.csv') + glob.glob(out + '/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print('    %-40s mean %16.1f  (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
tail -3 $out/summary.txt
for p in $out/p*/; do rm -rf $p; done
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/mftrace -o t -- python3 scratch/mf_pmc.py "$@" > $out/mftrace.log 2>&1
find $out/mftrace -type f ! -name '*kernel_trace.csv' -delete
echo "ta batch done"
