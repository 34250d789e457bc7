"""One replayed training step out of a rocprofv3 kernel trace: launches, wall, per-kernel totals.
usage: python3 scratch/step_table.py <dir> [top]"""
import csv, re, glob, collections, sys
f = (glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [re.sub(r"\(.*", "", r['Kernel_Name']).replace('void ', '') for r in rows]
sg = [i for i, n in enumerate(names) if 'sgd' in n]
which = int(sys.argv[3]) if len(sys.argv) > 3 else len(sg) // 2      # a replay in the middle of the timed loop (the tail of bench.py runs eager steps)
a, b = sg[which - 1] + 1, sg[which] + 1
step, sn = rows[a:b], names[a:b]
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
print('one replayed step: %d launches, wall %.2f ms, sum of kernel durations %.2f ms' % (len(step), (t1 - t0) / 1e6, busy / 1e6))
c = collections.Counter(sn); d = collections.defaultdict(float)
for r, n in zip(step, sn):
    d[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for n, k in sorted(c.items(), key=lambda kv: -d[kv[0]])[:top]:
    print('%-72s x%3d  %8.1f us total  %6.1f avg' % (n[:72], k, d[n], d[n] / k))

# the launches in order (a file next to the trace directory): index, start offset, duration, grid / workgroup size, name
try:
    with open(sys.argv[1].rstrip('/') + '_sequence.txt', 'w') as fh:
        for i, (r, n) in enumerate(zip(step, sn)):
            gx = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) * max(int(r.get('Grid_Size_Y', 1) or 1), 1) * max(int(r.get('Grid_Size_Z', 1) or 1), 1)
            wx = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 0)) or 0) * max(int(r.get('Workgroup_Size_Y', 1) or 1), 1) * max(int(r.get('Workgroup_Size_Z', 1) or 1), 1)
            fh.write('%3d  +%8.1f us  %6.1f us  wg %6d x %4d  vgpr %3s lds %6s  %s\n' % (i, (int(r['Start_Timestamp']) - t0) / 1e3,
                     (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, gx // max(wx, 1), wx, r.get('VGPR_Count', '?'), r.get('LDS_Block_Size', '?'), n[:110]))
except Exception as e:
    print('sequence not written:', e)
