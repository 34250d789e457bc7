cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/step_pmc
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  for n in 2 5; do
    timeout -k 10 280 rocprofv3 --pmc $c --output-format csv -d $out/${c}_$n -- python3 scratch/step_pmc.py $n > $out/${c}_$n.log 2>&1 || echo "$c $n failed"
  done
done
python3 - <<'PY' | tee gpurun_out/step_pmc/summary.txt
import csv, glob, collections, re
def total(d):
    per = collections.Counter(); n = 0
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            per[re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:60]] += float(r['Counter_Value']); n += 1
    return per, n
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    a, na = total('gpurun_out/step_pmc/%s_2' % c); b, nb = total('gpurun_out/step_pmc/%s_5' % c)
    per = {k: (b[k] - a.get(k, 0.0)) / 3.0 for k in b}
    res[c] = per
    print('%s: %d dispatches in 3 steps -> %.0f per step; sum %.1f MiB-units per step' % (c, nb - na, (nb - na) / 3.0, sum(per.values()) / 1024))
keys = set(res['FETCH_SIZE']) | set(res['WRITE_SIZE'])
rows = sorted(((2 * res['FETCH_SIZE'].get(k, 0) + res['WRITE_SIZE'].get(k, 0)) * 1024 / 1e6, k) for k in keys)      # KiB -> MB, gfx950: FETCH x 2
tot = sum(r[0] for r in rows)
print('HBM-side traffic per training step (2 x FETCH_SIZE + WRITE_SIZE, KiB): %.1f MB' % tot)
for mb, k in rows[::-1][:40]:
    print('  %8.1f MB  %s' % (mb, k))
PY
