import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn as nn
import _seeded as S
import crfconv_amd
from crfconv_amd import models, ops
from gpu_util import t, grads, DEV
B, N = 2, 4096
def collate(seed):
    pos = np.stack([S.make_cloud(seed + b, N, box=(2, 2, 1)) for b in range(B)])
    feats = np.concatenate([pos, S.uniform(seed, 'rgb', (B, N, 3), 0, 1)], -1)
    labels = S.integers(seed, 'y', (B, N), 0, 14)
    g = torch.Generator().manual_seed(seed)
    return crfconv_amd.multiscale_compute(t(pos), x=t(feats), y=t(labels), generator=g)
net = models.PointConvBig(6, 13, True, 3)
net.load_state_dict(S.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 4))
net = net.to(DEV).train(); net.classifier[1] = nn.Identity()
def run(data):
    for p in net.parameters(): p.grad = None
    for mod in net.modules():
        if isinstance(mod, nn.BatchNorm1d): mod.reset_running_stats()
    logits = net(data); loss = ops.training_loss(logits, data.y, None, ignore_index=-1); loss.backward()
    return logits.detach().clone(), float(loss.detach()), {k: v.clone() for k, v in grads(net).items()}
a = collate(100); run(a)
r1 = run(collate(200)); r2 = run(collate(200))
print('two fresh collates of the same batch: logits maxdiff %.3e, loss %r %r' % (float((r1[0]-r2[0]).abs().max()), r1[1], r2[1]))
c = collate(200)
for i,(m1,m2) in enumerate(zip(c.multiscale, collate(200).multiscale)):
    print(i, [bool(torch.equal(getattr(m1,k), getattr(m2,k))) for k in ('pos','neighbor_idx','sub_idx','up_idx')])
a.load_(c)
for i,(m1,m2) in enumerate(zip(a.multiscale, c.multiscale)):
    print('loaded', i, [bool(torch.equal(getattr(m1,k), getattr(m2,k))) for k in ('pos','neighbor_idx','sub_idx','up_idx')])
g = run(a)
print('load_: logits maxdiff %.3e, loss %r %r' % (float((g[0]-r1[0]).abs().max()), g[1], r1[1]))
worst = max(((float((g[2][k]-r1[2][k]).abs().max()), k) for k in r1[2]))
print('worst grad diff', worst)
