"""Sequence of one replayed step: every framework (at:: / rocclr / Cijk) launch with its two neighbours on either side.
usage: python3 scratch/step_seq.py <dir> <which>"""
import csv, re, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].replace('void ', '') for r in rows]
sg = [i for i, n in enumerate(names) if 'sgd' in n]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(sg) // 2
a, b = sg[which - 1] + 1, sg[which] + 1
short = lambda n: re.sub(r"\(.*", "", n)[:60]
for i in range(a, b):
    n = names[i]
    if n.startswith('at::') or 'rocclr' in n or n.startswith('Cijk'):
        dur = (int(rows[i]['End_Timestamp']) - int(rows[i]['Start_Timestamp'])) / 1e3
        print('%4d %5.1f us  %s' % (i - a, dur, n[:230]))
        print('        before: %s | %s' % (short(names[i - 2]), short(names[i - 1])))
        print('        after : %s | %s' % (short(names[i + 1]), short(names[min(i + 2, b - 1)])))
