#!/bin/bash
# usage: scratch/pmc_mfma.sh <outdir>  -- MFMA counters of every matrix-pipe kernel of the training step (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 250 rocprofv3 --pmc $set --kernel-include-regex "head_|wgrad_jobs_any_kernel|linear_fwd_kernel|wgrad_kernel|gemm_kernel|gemm_jobs_kernel|gemm_stats_jobs_kernel|gemm_pro_jobs_kernel|mlp_bwd_p1_kernel|mlp_small_fwd_kernel|wide_params_any_kernel|uvstats_mfma_kernel|bwd_params_kernel|bwd_edge_all_kernel" --output-format csv -d $out/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --graph 0 > $out.p$i.log 2>&1 || echo "pass $i failed/timeout"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '') + ' grid=%s' % r.get('Grid_Size', '?')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print('    %-34s mean %14.1f  (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
