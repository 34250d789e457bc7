import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import crfconv_amd
from crfconv_amd import ops
from crfconv_amd.models.point_conv_big import ResNetBBlock
from crfconv_amd.models import common
import _seeded as S
DEV = 'cuda:0'
B, N = 2, 4096
pos = np.stack([S.make_cloud(310 + b, N, box=(2, 2, 1)) for b in range(B)])
data = crfconv_amd.multiscale_compute(torch.from_numpy(pos).float().to(DEV), generator=torch.Generator().manual_seed(3))
lvl = data.multiscale[0]
blk = ResNetBBlock(32, 32).to(DEV).train()
x = torch.randn(B, N, 32, device=DEV, requires_grad=True)
xin = x * 1.0
h, al = common.mlp_fork(blk.lin_in, xin)
print('fork:', h.grad_fn, al.grad_fn, al is xin)
out = blk(xin, lvl.pos, lvl.neighbor_idx)
seen, stack = set(), [out.grad_fn]
while stack:
    f = stack.pop()
    if f is None or f in seen: continue
    seen.add(f); stack.extend(g for g, _ in f.next_functions)
mul = [f for f in seen if f.name() == 'MulBackward0'][0]
for f in seen:
    for g, i in f.next_functions:
        if g is mul: print('user', f.name(), [n.name() if n else None for n, _ in f.next_functions])
        if g is not None and 'MLP' in g.name(): print('edge', f.name(), '->', g.name(), i)
