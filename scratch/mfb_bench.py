"""Mean-field backward on the bench's level-0 tables: restructured (T + 3 launches) vs step-by-step launches.
Parity between the two and against float64 torch on a sample, event timing of backward alone (forward outside)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from crfconv_amd import ops
from crfconv_amd.graph import table_of

dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
B, N = int(os.environ.get('B', 4)), int(os.environ.get('N', 40960))
H, T = int(os.environ.get('H', 8)), int(os.environ.get('T', 3))
data, _ = bench.make_batch(0, B, N, dev, gen, 'morton')
ms0 = data.multiscale[0]
K = ms0.neighbor_idx.shape[2]
m = B * N
tab = table_of(ms0.neighbor_idx, N)
tab.reverse
g = torch.Generator().manual_seed(1)
z = torch.randn(m, H, generator=g).to(dev).requires_grad_()
y = (0.5 * torch.randn(m, H, generator=g)).to(dev).requires_grad_()
c = (torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)).to(dev).requires_grad_()
gout = torch.randn(m, H, generator=g).to(dev)


def run(old):
    ops._OLD_BWD_ENV = old
    for t in (z, y, c):
        t.grad = None
    out = ops.crf_meanfield(z, y, c, tab, T)
    out.backward(gout)
    return [t.grad.clone() for t in (z, y, c)]


new, ref = run(False), run(True)
for name, a, b in zip(('dz', 'dy', 'dc'), new, ref):
    print('%s: max |new - old| = %.3e  (max |old| %.3e)' % (name, float((a - b).abs().max()), float(b.abs().max())), flush=True)


def time_bwd(old, n=50):
    ops._OLD_BWD_ENV = old
    ts = []
    for it in range(n + 5):
        for t in (z, y, c):
            t.grad = None
        out = ops.crf_meanfield(z, y, c, tab, T)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out.backward(gout)
        b.record()
        torch.cuda.synchronize()
        if it >= 5:
            ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


for old in (True, False, True, False):
    med, lo = time_bwd(old)
    print('%-28s backward (eager, incl. host launch overhead): median %7.1f us  min %7.1f us' % (
        'step-by-step launches' if old else 'restructured (T+3 launches)', med, lo), flush=True)
