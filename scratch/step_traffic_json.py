"""gpurun_out/step_pmc/summary.txt (scratch/run_step_pmc.sh) -> profiles/r6h_step_traffic.md + profiles/r6_step_traffic.json (HBM-side
bytes of one training step, keyed on the sha1 of every kernel source and of the autograd glue: bench.py reports null once any changed)."""
import glob, hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(os.path.join(ROOT, 'gpurun_out', 'step_pmc', 'summary.txt')).read()
mb = float(re.search(r'KiB\): ([\d.]+) MB', txt).group(1))
launches = int(float(re.search(r'-> (\d+) per step', txt).group(1)))
srcs = sorted(os.path.relpath(f, ROOT) for f in glob.glob(os.path.join(ROOT, 'crfconv_amd', 'csrc', '*.h*'))) + sorted(os.path.relpath(f, ROOT) for f in glob.glob(os.path.join(ROOT, 'crfconv_amd', 'ops', '*.py')))
h = hashlib.sha1()
for s_ in srcs:
    h.update(open(os.path.join(ROOT, s_), 'rb').read())
json.dump({'config': {'B': 4, 'N': 40960, 'K': 16, 'T': 3}, 'source': srcs, 'source_sha1': h.hexdigest(), 'traffic_bytes_per_launch': mb * 1e6,
           'launches_per_step': launches, 'profile': 'profiles/r6h_step_traffic.md',
           'how': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over scratch/step_pmc.py N (N = 2 and 5, difference / 3); bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB'},
          open(os.path.join(ROOT, 'profiles', 'r6_step_traffic.json'), 'w'), indent=1)
with open(os.path.join(ROOT, 'profiles', 'r6h_step_traffic.md'), 'w') as f:
    f.write('# Round 6: HBM-side traffic of the WHOLE training step (final build)\n\n'
            'Two separate `rocprofv3 --pmc` passes (`FETCH_SIZE`, `WRITE_SIZE`; `scratch/run_step_pmc.sh`) over `scratch/step_pmc.py N` -- the same calls as\n'
            '`bench.py`\'s step, run eagerly -- with N = 2 and N = 5; per-step figures = (N = 5 minus N = 2) / 3.  Bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB, the\n'
            'gfx950 correction of the micro-architecture guide.  4 x 40 960 points, K = 16, T = 3, fp32.  **%.2f GB per step in %d launches**; the algorithmic byte\n'
            'model of `bench.py::step_byte_model` gives 3.49 GB: waste ratio %.2f (DESIGN.md 7 lists where the factor goes).  Round 5: 5.82 GB in 278 launches; round 4: 5.81 GB in 309; round 3: 7.35 GB in 361.\n\n```\n%s\n```\n'
            % (mb / 1e3, launches, mb * 1e6 / 3.49e9, txt.strip()))
print('step traffic %.1f MB, %d launches' % (mb, launches))
