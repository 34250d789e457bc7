"""Mean-field forward / backward per decoder level (bench.roofline_meanfield*), for A/B runs of library variants
(CRFCONV_LIB=scratch/variants/lib_<name>.so) and of the one-launch forward where it is supported."""
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
        lib = _lib.load()
        tab, m, K, z, y, Q, P, _ = bench._meanfield_problem(data, dev, H, level=level)
        if lib.crfconv_meanfield_fused_supported(m, H, K, 1, T) == 1:
            wsb = lib.crfconv_meanfield_fused_workspace()
            ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
            s = torch.empty(m, K, device=dev); xs = torch.empty(T, m, H, device=dev)
            st = stream_ptr()
            def launch_fused():
                _lib.call('crfconv_meanfield_forward_fused', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src,
                          K, 1, m, H, ptr(Q), ptr(P), T, ptr(s), ptr(xs), ptr(ws), wsb, st)
            line += '   fused one-launch %6.2f us' % (bench._event_time(launch_fused)[0] * 1e6)
        print(line, flush=True)
