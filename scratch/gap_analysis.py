"""Busy vs idle time of the GPU inside the timed steps of a kernel trace (graph-replay mode)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# steps are delimited by the optimizer's multi_tensor_apply kernels; take the last N*per_step launches
names = [r[2] for r in rows]
marks = [i for i, n in enumerate(names) if 'ce_fwd_kernel' in n]
print('steps found', len(marks))
if len(marks) > 6:
    a, b = marks[-6], marks[-1]
    seg = rows[a:b]
    wall = seg[-1][0] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    pos = [g for g in gaps if g > 0]
    print('5 steps: wall %.3f ms/step, kernel-busy %.3f ms/step, launches/step %d' % (wall / 5e6, busy / 5e6, len(seg) / 5))
    print('idle between kernels %.3f ms/step (mean gap %.2f us, overlapped pairs %d)' % (sum(pos) / 5e6, sum(pos) / len(pos) / 1e3, sum(1 for g in gaps if g <= 0)))
    import collections
    h = collections.Counter(min(int(g / 1000), 10) for g in pos)
    print('gap histogram (us: count/step):', {k: v // 5 for k, v in sorted(h.items())})
