"""PointConv forward + backward in train mode at level 0 (d = 8), 1 (d = 16), 3 (d = 64) and 4 (d = 128), 5 eager calls each (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from crfconv_amd import ops
from crfconv_amd.graph import table_of
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
for lvl, d in ((0, 8), (1, 16), (3, 64), (4, 128)):
    ms = data.multiscale[lvl]
    B, N, K = ms.neighbor_idx.shape
    tab = table_of(ms.neighbor_idx, N); tab.reverse
    pos = ms.pos.reshape(-1, 3).contiguous()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B * N, d, generator=g).to(dev).requires_grad_()
    W1 = (0.5 * torch.randn(d, 3, generator=g)).to(dev).requires_grad_()
    W2 = (0.5 * torch.randn(d, d, generator=g)).to(dev).requires_grad_()
    bn1, bn2 = torch.nn.BatchNorm1d(d).to(dev), torch.nn.BatchNorm1d(d).to(dev)
    gout = torch.randn(B * N, d, generator=g).to(dev)
    mom = ops.relpos_moments(pos, pos, tab)
    for it in range(5):
        for t in (x, W1, W2):
            t.grad = None
        ops.point_conv(x, pos, None, tab, W1, bn1, W2, bn2, True, moments=mom).backward(gout)
torch.cuda.synchronize()
print('done')
