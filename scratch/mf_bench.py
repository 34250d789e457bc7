"""Micro-benchmark of the level-0 mean-field forward/backward (for rocprofv3)."""
import sys, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
import crfconv_amd
from crfconv_amd import ops
from crfconv_amd.graph import table_of
sort = sys.argv[1] if len(sys.argv) > 1 else 'morton'
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, sort)
print(json.dumps(bench.roofline_meanfield(data, dev, 8, 3, iters=int(sys.argv[3]) if len(sys.argv) > 3 else 100)))
if len(sys.argv) > 2 and sys.argv[2] == 'bwd':
    ms0 = data.multiscale[0]
    tab = table_of(ms0.neighbor_idx, 40960)
    m, H = 4 * 40960, 8
    z = torch.randn(m, H, device=dev, requires_grad=True); y = torch.randn(m, H, device=dev, requires_grad=True)
    c = (torch.eye(H) + 0.1 * torch.randn(H, H)).to(dev).requires_grad_(True)
    g = torch.randn(m, H, device=dev)
    for _ in range(20):
        out = ops.crf_meanfield(z, y, c, tab, 3)
        out.backward(g)
    torch.cuda.synchronize()
