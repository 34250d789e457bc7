#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_pc
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-include-regex "uvstats|bwd_params|bwd_input|uv_combine|reduce_partials|uv_bwd_reduce|fold2|bwd_dump|a1_reduce|wide_params" --output-format csv -d gpurun_out/pmc_pc/p$i -- python3 scratch/pc_pmc.py > gpurun_out/pmc_pc.p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY' | tee gpurun_out/pmc_pc_summary.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_pc/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print('    %-34s mean %14.1f  (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
