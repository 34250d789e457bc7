"""profiles/r6_meanfield_ta.md from the counter summary of scratch/run_r6_ta.sh (TA / TCP / TCC / SQ passes over scratch/mf_pmc.py) and the kernel
trace of the same batch: per kernel and per CU, what the vector-memory path did -- line accesses of the vector cache against the launch's
cycles, busy fractions, stall counters.  usage: python3 scratch/ta_table.py gpurun_out/<tag>"""
import collections, csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
txt = open(os.path.join(d, 'summary.txt')).read()
blocks = {b.split('\n')[0].strip(): dict((k, float(v)) for k, v in re.findall(r'(\w+)\s+mean\s+([\d.]+)', b))
          for b in re.split(r'\n(?=crf::)', '\n' + txt) if b.strip().startswith('crf::')}
f = (glob.glob(d + '/mftrace/*/*kernel_trace.csv') + glob.glob(d + '/mftrace/*kernel_trace.csv'))[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    dur[re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
CUS = 256
out = []
out.append('# Round 6: what the vector-memory path does in the level-0 mean-field kernels (TA / TCP / TCC / SQ counters), MI355X\n')
out.append('`bash scratch/run_r6_ta.sh <tag> 20` = one `rocprofv3 --pmc <set>` pass per counter set (never combined with a trace) over `scratch/mf_pmc.py 20` '
           '(level 0: m = 163 840, H = 8, K = 16, T = 3; 20 forward + backward calls in each forward form), then a `--kernel-trace` pass of the same script '
           'for the durations.  Raw means: `profiles/r6_meanfield_ta_counters.txt`.  Per-CU figures divide the device sums by %d CUs; "cycles" = average '
           'duration x 2.1 GHz (the clock under a profiled load is 1.9-2.4 GHz, MI355X_MICROARCH.md: ratios to it carry +-12 %%).\n' % CUS)
out.append('| kernel | avg us | vector-cache line accesses / CU | accesses per cycle and CU | TA busy (avg / max over CUs) of the launch | TCP -> L2 read requests / CU | L2 hit rate | '
           'TCP pending-stall / tag-conflict-stall share of TCP-active cycles | VALU instr / wave | LDS bank-conflict cycles per LDS instr |\n|---|---|---|---|---|---|---|---|---|---|')
for k in sorted(blocks):
    b = blocks[k]
    if k not in dur or 'TCP_TOTAL_CACHE_ACCESSES_sum' not in b:
        continue
    us = sum(dur[k]) / len(dur[k])
    cyc = us * 2100.0
    acc = b['TCP_TOTAL_CACHE_ACCESSES_sum'] / CUS
    gate = b.get('TCP_GATE_EN1_sum', 0) / CUS
    out.append('| `%s` | %.2f | %.0f | **%.2f** | %.2f / %.2f | %.0f | %.2f | %.2f / %.2f | %.0f | %.2f |' % (
        k, us, acc, acc / cyc, b['TA_BUSY_avr'] / cyc, b['TA_BUSY_max'] / cyc, b['TCP_TCC_READ_REQ_sum'] / CUS,
        b['TCC_HIT_sum'] / max(1.0, b['TCC_HIT_sum'] + b['TCC_MISS_sum']),
        b.get('TCP_PENDING_STALL_CYCLES_sum', 0) / CUS / max(gate, 1), b.get('TCP_READ_TAGCONFLICT_STALL_CYCLES_sum', 0) / CUS / max(gate, 1),
        b['SQ_INSTS_VALU'] / b['SQ_WAVES'], b['SQ_LDS_BANK_CONFLICT'] / max(1.0, b['SQ_INSTS_LDS'])))
open(os.path.join(ROOT, 'profiles', 'r6_meanfield_ta_table.md'), 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
