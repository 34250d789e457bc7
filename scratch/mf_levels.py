"""Mean-field forward / backward per decoder level (bench.roofline_meanfield*), for A/B runs of library variants
(CRFCONV_LIB=scratch/variants/lib_<name>.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from crfconv_amd import _lib
from crfconv_amd.graph import ptr, stream_ptr
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
T = 3
print('lib', os.environ.get('CRFCONV_LIB', 'default'), flush=True)
for rep in range(2):
    for level in range(4):
        H = 8 << level
        f = bench.roofline_meanfield(data, dev, H, T, level=level)
        b = bench.roofline_meanfield_bwd(data, dev, H, T, level=level) if '--bwd' in sys.argv else None
        line = 'level %d H %2d fwd avg %6.2f min %6.2f us' % (level, H, f['avg_launch_us'], f['min_launch_us'])
        if b:
            line += '   bwd avg %6.2f min %6.2f us' % (b['avg_launch_us'], b['min_launch_us'])
        print(line, flush=True)
