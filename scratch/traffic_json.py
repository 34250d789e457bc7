"""Turns the summary of scratch/pmc.sh (level-0 mean-field passes of scratch/mf_pmc.py) into the committed provenance
files: profiles/r6_meanfield_traffic.json, profiles/r6_meanfield_bwd_traffic.json (HBM-side bytes per launch + sha1 of the
kernel source they were measured on) and profiles/r6a_meanfield_pmc.md (the raw counter means).
usage: python3 scratch/traffic_json.py gpurun_out/r5k/mfpmc_summary.txt"""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(sys.argv[1]).read()
blocks = {b.split('\n')[0].strip(): dict((k, float(v)) for k, v in re.findall(r'(\w+)\s+mean\s+([\d.]+)', b))
          for b in re.split(r'\n(?=crf::)', txt) if b.startswith('crf::')}


def hbm_bytes(name):           # FETCH_SIZE counts half the bytes on gfx950 (MI355X_MICROARCH.md); both in KiB
    d = blocks[name]
    return (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024


T = 3
FWD_FIRST, FWD_STEP = 'crf::sim_step_fast_kernel<8, 16, true, true>', 'crf::step_fast_kernel<8, 16, true>'
REV_FIRST, REV_CHAIN, REV_FINAL = ('crf::bwd_rev_kernel<8, 2, 3, 0, true, 4, true>', 'crf::bwd_rev_kernel<8, 2, 3, 0, false, 4, true>',
                                   'crf::bwd_rev_kernel<8, 2, 2, 1, false, 4, true>')
EDGE = 'crf::bwd_edge_all_kernel<8, 16, true>'
FWD_BLOCK = 'crf::mf_block_kernel<8, 16, 8, 5, true, false>'
fwd = hbm_bytes(FWD_BLOCK)                     # the product's forward on a local table (round 6): ONE launch
fwd_steps = hbm_bytes(FWD_FIRST) + (T - 1) * hbm_bytes(FWD_STEP)
bwd = hbm_bytes(REV_FIRST) + (T - 2) * hbm_bytes(REV_CHAIN) + hbm_bytes(EDGE) + hbm_bytes(REV_FINAL)
fsrc = ['crfconv_amd/csrc/crf_block.hip', 'crfconv_amd/csrc/crf.hip', 'crfconv_amd/csrc/crf_common.hpp']
bsrc = ['crfconv_amd/csrc/crf_bwd.hip', 'crfconv_amd/csrc/crf_common.hpp']


def sha_of(srcs):
    h = hashlib.sha1()
    for src in srcs:
        h.update(open(os.path.join(ROOT, src), 'rb').read())
    return h.hexdigest()


how = ('rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (scratch/pmc.sh, scratch/mf_pmc.py, scratch/traffic_json.py); '
       'bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per the gfx950 correction of MI355X_MICROARCH.md; ')
cfg = {'m': 163840, 'H': 8, 'K': 16, 'T': T, 'u16': True}
for name, val, srcs, what in (('r6_meanfield_traffic.json', fwd, fsrc, 'mf_block_kernel (one launch; the per-step launches of the same problem: %.1f MB)' % (fwd_steps / 1e6)),
                              ('r6_meanfield_bwd_traffic.json', bwd, bsrc, 'bwd_rev<chain, first> + bwd_rev<chain> + bwd_edge_all + bwd_rev<final>')):
    json.dump({'config': cfg, 'source': srcs, 'source_sha1': sha_of(srcs), 'traffic_bytes_per_launch': val,
               'profile': 'profiles/r6a_meanfield_pmc.md', 'how': how + what}, open(os.path.join(ROOT, 'profiles', name), 'w'), indent=1)
    print(name, '%.1f MB' % (val / 1e6))
with open(os.path.join(ROOT, 'profiles', 'r6a_meanfield_pmc.md'), 'w') as f:
    f.write('# rocprofv3 --pmc passes, level-0 mean-field forward + backward (m = 163840, H = 8, K = 16, T = 3), MI355X, round 6\n'
            '# scratch/pmc.sh <outdir> <regex> scratch/mf_pmc.py ; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE counts half the bytes on gfx950;\n'
            '# SQ_* counters as rocprofv3 reports them (counter units not converted).  sha1 forward sources %s, backward sources %s\n'
            '# forward HBM-side bytes per call = %.1f MB as one launch (mf_block_kernel), %.1f MB as three (sim_step_fast + 2 x step_fast); backward = %.1f MB (algorithmic: 46.5 MB / 104.2 MB)\n\n' % (sha_of(fsrc), sha_of(bsrc), fwd / 1e6, fwd_steps / 1e6, bwd / 1e6))
    f.write('| kernel | waves | SQ_WAVE_CYCLES / SQ_WAVES | waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | vector-memory reads | fetched MB | written MB |\n|---|---|---|---|---|---|---|\n')
    for k in sorted(blocks):
        d = blocks[k]
        f.write('| `%s` | %d | %.0f | %.0f %% | %d | %.1f | %.1f |\n' % (
            k, d['SQ_WAVES'], d['SQ_WAVE_CYCLES'] / d['SQ_WAVES'], 100 * d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES'], d['SQ_INSTS_VMEM_RD'],
            2 * d['FETCH_SIZE'] * 1024 / 1e6, d['WRITE_SIZE'] * 1024 / 1e6))
    f.write('\n```\n' + txt.strip() + '\n```\n')
