"""profiles/r6_meanfield_rocprof.json + profiles/r6a_meanfield_kernel_stats.md from a rocprofv3 --kernel-trace run of scratch/mf_pmc.py (the
level-0 mean-field forward + backward, m = 163840, H = 8, K = 16, T = 3): per-kernel AVERAGE durations, their sums per direction, and the
sha1 of the kernel sources (bench.py reports `frac_rocprof` only while the sources are unchanged).
usage: python3 scratch/mf_rocprof_json.py <trace dir>"""
import collections, csv, glob, hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = (glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
    dur[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
avg = {k: sum(v) / len(v) for k, v in dur.items()}
T = 3
FWD_STEPS = [('crf::sim_step_fast_kernel<8, 16, true, true>', 1), ('crf::step_fast_kernel<8, 16, true>', T - 1)]
FWD = [('crf::mf_block_kernel<8, 16, 8, 5, true, false>', 1)]       # the product's forward on a local table (round 6): one launch
BWD = [('crf::bwd_rev_kernel<8, 2, 3, 0, true, 4, true>', 1), ('crf::bwd_rev_kernel<8, 2, 3, 0, false, 4, true>', T - 2), ('crf::bwd_edge_all_kernel<8, 16, true>', 1),
       ('crf::bwd_rev_kernel<8, 2, 2, 1, false, 4, true>', 1)]
fwd = sum(avg[k] * n for k, n in FWD)
fwd_steps = sum(avg[k] * n for k, n in FWD_STEPS)
bwd = sum(avg[k] * n for k, n in BWD)
srcs = ['crfconv_amd/csrc/crf.hip', 'crfconv_amd/csrc/crf_block.hip', 'crfconv_amd/csrc/crf_bwd.hip', 'crfconv_amd/csrc/crf_common.hpp']
h = hashlib.sha1()
for s in srcs:
    h.update(open(os.path.join(ROOT, s), 'rb').read())
m, H, K = 163840, 8, 16
alg_f = m * (4 * (K - 1) + 4 * H * (2 * T + 1))
alg_b = m * (2 * (4 * (K - 1) + 4 * H * (2 * T + 1)) + 4 * K + 4)
json.dump({'config': {'m': m, 'H': H, 'K': K, 'T': T}, 'source': srcs, 'source_sha1': h.hexdigest(), 'fwd_us': fwd, 'fwd_steps_us': fwd_steps, 'bwd_us': bwd,
           'kernels_us': {k: avg[k] for k, _ in FWD + FWD_STEPS + BWD}, 'calls': {k: len(dur[k]) for k, _ in FWD + FWD_STEPS + BWD},
           'profile': 'profiles/r6a_meanfield_kernel_stats.md',
           'how': 'rocprofv3 --kernel-trace over scratch/mf_pmc.py (N forward + backward calls in each forward form); average End - Start per kernel; '
                  'forward = mf_block_kernel (one launch), forward per step = sim_step_fast + 2 x step_fast, backward = bwd_rev<chain, first> + bwd_rev<chain> + bwd_edge_all + bwd_rev<final>'},
          open(os.path.join(ROOT, 'profiles', 'r6_meanfield_rocprof.json'), 'w'), indent=1)
with open(os.path.join(ROOT, 'profiles', 'r6a_meanfield_kernel_stats.md'), 'w') as o:
    o.write('# Round 6: rocprofv3 --kernel-trace, level-0 mean-field forward + backward alone (m = 163840, H = 8, K = 16, T = 3), MI355X\n\n'
            '`rocprofv3 --kernel-trace --output-format csv -- python3 scratch/mf_pmc.py %d` (calls per forward form; the backward is the same in both).' % (len(dur[FWD[0][0]])) + '  Algorithmic bytes (SURVEY 8(d)): forward %.1f MB, '
            'backward %.1f MB; peak 8 TB/s.\n\n| kernel | calls | avg us |\n|---|---|---|\n' % (alg_f / 1e6, alg_b / 1e6))
    for k, n in FWD + FWD_STEPS + BWD:
        o.write('| `%s` | %d | %.2f |\n' % (k, len(dur[k]), avg[k]))
    o.write('\n| | sum of averages (us) | fraction of the HBM roofline on algorithmic bytes |\n|---|---|---|\n')
    o.write('| forward, block-resident rows (1 launch: the product\'s path on this table) | %.2f | %.3f |\n| forward, one launch per step (3 launches) | %.2f | %.3f |\n| backward (4 launches) | %.2f | %.3f |\n'
            % (fwd, alg_f / (fwd * 1e-6) / 8e12, fwd_steps, alg_f / (fwd_steps * 1e-6) / 8e12, bwd, alg_b / (bwd * 1e-6) / 8e12))
    o.write('\nsha1 of %s: %s (bench.py `frac_rocprof`).  The HIP-event figure of bench.py (`frac`) is ~1 us per dispatch lower: the profiler\'s '
            'own overhead per kernel.\n' % (', '.join(srcs), h.hexdigest()))
print('fwd %.2f us (per step: %.2f us) bwd %.2f us' % (fwd, fwd_steps, bwd))
