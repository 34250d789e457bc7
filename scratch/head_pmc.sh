#!/bin/bash
# matrix-pipe counters of the classifier-head kernels (separate --pmc passes, no trace): busy cycles per MFMA instruction, utilisation
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/head_pmc
mkdir -p $out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --pmc $set --kernel-include-regex "head_" --output-format csv -d $out/p$i -- python3 scratch/head_bench.py recompute > $out.p$i.log 2>&1 || echo "pass $i failed/timeout"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    m = {c: sum(v) / len(v) for c, v in agg[k].items()}
    print(k)
    for c in sorted(m):
        print('    %-34s %16.1f' % (c, m[c]))
    if m.get('SQ_INSTS_MFMA'):
        print('    -> MFMA busy cycles per MFMA instruction (per SIMD): %.1f' % (m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / m['SQ_INSTS_MFMA']))
        print('    -> MFMA busy / (SQ_BUSY_CYCLES / 32 SEs x 1024 SIMDs): %.3f' % (m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (m.get('SQ_BUSY_CYCLES', 1) / 32 * 1024)))
PY
rm -rf $out/p*
