import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crfconv_amd
from crfconv_amd import ops
from crfconv_amd.graph import NeighborTable
dev = 'cuda'
g = torch.Generator().manual_seed(0)
res = {}
for mode in ('now', 'late'):
    out = []
    cs = [torch.nn.Parameter((torch.randn(H, H, generator=torch.Generator().manual_seed(H)) * 0.1).to(dev)) for H in (16, 32, 64)]
    tabs, zs, ys = [], [], []
    for H, n in zip((16, 32, 64), (4096, 2560, 640)):
        gg = torch.Generator().manual_seed(n)
        idx = torch.randint(0, n, (1, n, 16), generator=gg); idx[0, :, 0] = torch.arange(n)
        tabs.append(NeighborTable(idx.to(dev), n))
        zs.append(torch.randn(n, H, generator=gg).to(dev).requires_grad_(True)); ys.append(torch.randn(n, H, generator=gg).to(dev).requires_grad_(True))
    def run():
        mats = ops.crf_matrices_batched(cs)
        loss = 0
        for c, mat, t, z, y in zip(cs, mats, tabs, zs, ys):
            o = ops.crf_meanfield(z, y, c, t, 3, k0=1, matrices=mat)
            loss = loss + (o * torch.linspace(0, 1, o.numel(), device=dev).reshape(o.shape)).sum()
        return loss
    if mode == 'late':
        with ops.deferred_weight_grads():
            run().backward()
    else:
        run().backward()
    torch.cuda.synchronize()
    res[mode] = [c.grad.clone() for c in cs] + [z.grad.clone() for z in zs]
for a, b, name in zip(res['now'], res['late'], ['dc16', 'dc32', 'dc64', 'dz16', 'dz32', 'dz64']):
    print(name, float((a - b).abs().max()), float(a.abs().max()))
