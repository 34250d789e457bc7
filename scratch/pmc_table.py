"""Counter summaries of scratch/pmc.sh / pmc_mfma.sh / run_pcpmc.sh (blocks "kernel\\n    COUNTER mean v (n=..)") -> a markdown table.
usage: python3 scratch/pmc_table.py <summary.txt> <out.md> <title> [--pointconv-json]
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES): the busy counter counts SIMD cycles (32 per v_mfma_f32_16x16x4_f32),
SQ_BUSY_CYCLES is summed over the 32 shader engines of the 8 XCDs, and there are 1024 SIMDs = 32 x 32."""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, out, title = sys.argv[1], sys.argv[2], sys.argv[3]
txt = open(src).read()
blocks = {}
for b in re.split(r'\n(?=crf::)', txt[txt.index('crf::'):]):
    name = b.split('\n')[0].strip()
    blocks[name] = dict((k, (float(v), int(n))) for k, v, n in re.findall(r'(\w+)\s+mean\s+([\d.]+)\s+\(n=(\d+)\)', b))
rows = []
for name in sorted(blocks):
    d = {k: v[0] for k, v in blocks[name].items()}
    n = max(v[1] for v in blocks[name].values())
    busy = d.get('SQ_BUSY_CYCLES', 0.0)
    mfma, mb = d.get('SQ_INSTS_MFMA', 0.0), d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    waves = d.get('SQ_WAVES', 0.0)
    valu = d.get('SQ_INSTS_VALU', 0.0)
    wc = d.get('SQ_WAVE_CYCLES', 0.0)
    wait = d.get('SQ_WAIT_ANY')
    rows.append('| `%s` | %d | %d | %s | %s | %s | %s | %s | %.1f | %.1f |' % (
        name, n, waves, '%d' % mfma if mfma else '0', ('%.3f' % (mb / (32.0 * busy))) if (mb and busy) else '-',
        '%d' % (valu / waves) if waves and valu else '-', ('%.0f %%' % (100 * wait / wc)) if (wait and wc) else '-',
        ('%.0f' % (busy / 32.0)) if busy else '-', 2 * d.get('FETCH_SIZE', 0.0) * 1024 / 1e6, d.get('WRITE_SIZE', 0.0) * 1024 / 1e6))
with open(out, 'w') as f:
    f.write('# %s\n\n' % title)
    f.write('Separate `rocprofv3 --pmc` passes per counter set (never combined with a trace), per-dispatch means.  MFMA utilisation = '
            '`SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)` (busy counter: SIMD cycles, 32 per `v_mfma_f32_16x16x4_f32`; `SQ_BUSY_CYCLES` is summed '
            'over 32 shader engines; 1024 SIMDs).  Fetched MB = 2 x `FETCH_SIZE` KiB (the gfx950 correction of MI355X_MICROARCH.md), written MB = `WRITE_SIZE` KiB.  '
            'Kernel cycles = `SQ_BUSY_CYCLES / 32`.\n\n')
    f.write('| kernel (grid = threads) | dispatches | waves | MFMA instr | MFMA utilisation | VALU instr / wave | waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | kernel cycles | fetched MB | written MB |\n|---|---|---|---|---|---|---|---|---|---|\n')
    f.write('\n'.join(rows) + '\n\n```\n' + txt[txt.index('crf::'):].strip() + '\n```\n')
print('wrote', out, len(rows), 'kernels')
if '--pointconv-json' in sys.argv:
    def hbm(k):
        d = blocks[k]
        return (2 * d['FETCH_SIZE'][0] + d['WRITE_SIZE'][0]) * 1024
    fwd = hbm('crf::uvstats_kernel<8>') + sum(hbm(k) for k in blocks if k.startswith('crf::uv_combine_kernel') or k.startswith('crf::reduce_partials_kernel')) * 0  # see note
    srcs = ['crfconv_amd/csrc/pointconv.hip']
    h = hashlib.sha1()
    for s_ in srcs:
        h.update(open(os.path.join(ROOT, s_), 'rb').read())
    json.dump({'config': {'m': 163840, 'd': 8, 'K': 16}, 'source': srcs, 'source_sha1': h.hexdigest(), 'traffic_bytes_per_launch': fwd,
               'profile': 'profiles/' + os.path.basename(out),
               'how': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (scratch/run_pcpmc.sh, scratch/pc_pmc.py, scratch/pmc_table.py); '
                      'bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB; the dominant forward kernel uvstats_kernel<8> (the combine / reduce launches of the '
                      'forward are shared names across widths in the counter files and are listed in the profile, not added here)'},
              open(os.path.join(ROOT, 'profiles', 'r6_pointconv_traffic.json'), 'w'), indent=1)
    print('pointconv uvstats<8> HBM-side bytes per launch: %.1f MB' % (fwd / 1e6))
