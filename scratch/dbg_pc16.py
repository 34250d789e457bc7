import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import _seeded as S
from crfconv_amd.models import PointConv
from oracle import crf_oracle as O
from oracle import native as onative
d = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B, N, K = 2, 200, 16
pos = np.stack([S.make_cloud(60 + d + b, N) for b in range(B)])
nbr = onative.oracle_knn_batch(pos, pos, K)
m = PointConv(d)
sd = S.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 31)
m.load_state_dict(sd)
x = S.uniform(d, 'x', (B, N, d)); gout = S.uniform(d, 'g', (B, N, d))
def oracle(dtype):
    prm = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    xr = torch.from_numpy(x).to(dtype).requires_grad_(True)
    ref = O.point_conv(prm, '', xr, torch.from_numpy(pos).to(dtype), torch.from_numpy(nbr), True)
    (ref * torch.from_numpy(gout).to(dtype)).sum().backward()
    return prm, xr, ref
prm64, xr64, ref64 = oracle(torch.float64)
m = m.cuda().train()
xd = torch.from_numpy(x).cuda().requires_grad_(True)
out = m(xd, torch.from_numpy(pos).cuda(), torch.from_numpy(nbr).cuda())
(out * torch.from_numpy(gout).cuda()).sum().backward()
print('out err', float((out.detach().cpu().double() - ref64.detach()).abs().max()))
for k, p in m.named_parameters():
    g = p.grad.cpu().double(); r = prm64[k].grad
    e = (g - r).abs()
    print('%-40s err %.3e  refmax %.3e  at %s' % (k, float(e.max()), float(r.abs().max()), np.unravel_index(int(e.argmax()), e.shape)))
k = 'weight_nn.0.lin.weight'
g = dict(m.named_parameters())[k].grad.cpu().double(); r = prm64[k].grad
print((g - r))
print('var1 check: pre-activation stats')
