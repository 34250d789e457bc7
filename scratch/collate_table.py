import sys, csv, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'cat2_kernel' in r['Kernel_Name']]
a, b = marks[-2], marks[-1]
seg = rows[a + 1:b]
agg = collections.OrderedDict()
for r in seg:
    n = r['Kernel_Name'].split('(')[0][:90]
    t = agg.setdefault(n, [0, 0.0])
    t[0] += 1; t[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
wall = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3 / 10
print('one collate + refresh replay: %d launches, wall %.1f us, sum of kernel durations %.1f us' % (len(seg) // 10, wall, sum(v[1] for v in agg.values()) / 10))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-90s x%3d %8.1f us total %7.1f avg' % (n, c // 10, t / 10, t / c))
