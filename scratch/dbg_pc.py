import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import _seeded as S
from crfconv_amd import ops
from crfconv_amd.graph import NeighborTable
from oracle import native as onative
torch.manual_seed(0)
B, N, K, d = 2, 200, 16, 8
pos = np.stack([S.make_cloud(60 + b, N) for b in range(B)])
nbr = onative.oracle_knn_batch(pos, pos, K)
x = torch.randn(B*N, d); A1 = torch.randn(d, 3); b1 = torch.randn(d)*0.3; W2 = torch.randn(d, d)*0.5
g2 = torch.rand(d)+0.5; be2 = torch.randn(d)*0.2; rm = torch.randn(d)*0.1; rv = torch.rand(d)+0.5
gout = torch.randn(B*N, d)
P = torch.from_numpy(pos).reshape(-1, 3)
gidx = (torch.from_numpy(nbr) + (torch.arange(B)*N).view(B,1,1)).reshape(B*N, K)
def ref(train):
    xs, A, b, W, g, be = [t.clone().requires_grad_(True) for t in (x, A1, b1, W2, g2, be2)]
    rel = P[:, None, :] - P[gidx]            # [M,K,3]
    pre = rel @ A.t() + b
    h1 = torch.nn.functional.leaky_relu(pre, 0.1)
    h2 = h1 @ W.t()
    if train:
        w = torch.nn.functional.batch_norm(h2.reshape(-1, d), None, None, g, be, True, 0.1, 1e-5).reshape(h2.shape)
    else:
        w = torch.nn.functional.batch_norm(h2.reshape(-1, d), rm, rv, g, be, False, 0.1, 1e-5).reshape(h2.shape)
    out = (w * xs[gidx]).sum(1)
    (out * gout).sum().backward()
    return out.detach(), [t.grad for t in (xs, A, b, W, g, be)]
tab = NeighborTable(torch.from_numpy(nbr).cuda(), N)
mean_rel, cov, n = ops.relpos_moments(P.cuda(), P.cuda(), tab)
for train in (False, True):
    ro, rg = ref(train)
    ins = [t.clone().cuda().requires_grad_(True) for t in (x, A1, b1, W2, g2, be2)]
    aux = {}
    out = ops._PointConv.apply(ins[0], ins[1], ins[2], ins[3], ins[4], ins[5], P.cuda(), P.cuda(), tab, mean_rel.float(), train, rm.cuda(), rv.cuda(), aux)
    (out * gout.cuda()).sum().backward()
    print('train', train, 'out err', float((out.cpu()-ro).abs().max()))
    for name, a, r in zip(('dx','dA1','db1','dW2','dg2','dbe2'), ins, rg):
        print('  ', name, 'err %.3e  ref max %.3e' % (float((a.grad.cpu()-r).abs().max()), float(r.abs().max())))
    if not train:
        print(ins[1].grad.cpu()[:3], rg[1][:3])
