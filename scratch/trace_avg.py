"""Average duration per kernel (names matching argv[2], a regex) of a rocprofv3 --kernel-trace csv under argv[1]; argv[3]: calls to skip per kernel."""
import collections, csv, glob, re, sys
f = (glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[0]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else '.')
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
    if pat.search(n):
        dur[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0.0
for k in sorted(dur):
    v = dur[k][skip:] or dur[k]
    print('%-64s calls %4d  avg %7.2f us  min %7.2f' % (k[:64], len(v), sum(v) / len(v), min(v)))
