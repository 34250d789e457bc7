#!/bin/bash
# usage: scratch/pmc.sh <outdir> <kernel-regex> <script args...>   -- separate, time-boxed rocprofv3 --pmc passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; shift
rx=$1; shift
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-include-regex "$rx" --output-format csv -d $out/p$i -- python3 "$@" > $out.p$i.log 2>&1 || echo "pass $i failed/timeout"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print('    %-34s mean %14.1f  (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
