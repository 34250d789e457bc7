import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np, torch
import _seeded as S
import crfconv_amd
from crfconv_amd import models, ops
dev = torch.device('cuda', 0)
B, N = 2, 2048
pos = np.stack([S.make_cloud(300 + b, N, box=(2, 2, 1)) for b in range(B)])
feats = np.concatenate([pos, S.uniform(300, 'rgb', (B, N, 3), 0, 1)], -1).astype(np.float32)
labels = S.integers(300, 'y', (B, N), 1, 14)
choices, n = [], N
for i, r in enumerate((4, 4, 4, 2, 2)):
    choices.append(torch.from_numpy(S.permutation(300, 'c%d' % i, n)[: n // r])); n //= r
def batch(sl):
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a[sl])).to(dev)
    return crfconv_amd.multiscale_compute(f(pos), x=f(feats), y=f(labels), choices=choices, ratio=(4, 4, 4, 2, 2))
for use_crf in (True, False):
    net = models.PointConvBig(6, 13, use_crf, 3)
    net.load_state_dict(S.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 8))
    net = net.to(dev).eval()
    def grads_of(data):
        for p in net.parameters(): p.grad = None
        logits = net(data)
        loss = ops.training_loss(logits, data.y, None, ignore_index=-1); loss.backward()
        return logits.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    lb, gb = grads_of(batch(slice(0, B)))
    l0, g0 = grads_of(batch(slice(0, 1))); l1, g1 = grads_of(batch(slice(1, 2)))
    print('use_crf', use_crf, 'logits shard vs big: %.3e %.3e' % (float((lb[:N] - l0).abs().max()), float((lb[N:] - l1).abs().max())))
    worst = sorted(((float(((g0[k] + g1[k]) / 2 - gb[k]).abs().max()) / max(1e-12, float(gb[k].abs().max())), k) for k in gb), reverse=True)[:8]
    for w in worst: print('   %.3e  %s' % w)

# ---- which module first differs between the big batch and a shard (forward hooks, eval mode)?
net = models.PointConvBig(6, 13, False, 3)
net.load_state_dict(S.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 8))
net = net.to(dev).eval()
outs = {}
def hook(name):
    def f(mod, inp, out):
        if torch.is_tensor(out): outs.setdefault(name, []).append(out.detach().clone())
    return f
for name, mod in net.named_modules():
    if name and name.count('.') <= 1: mod.register_forward_hook(hook(name))
with torch.no_grad():
    net(batch(slice(0, B))); net(batch(slice(1, 2)))
for name, (big, sh) in outs.items():
    if big.dim() == 3 and big.shape[0] == B:
        d = float((big[1] - sh[0]).abs().max())
    elif big.dim() == 2 and big.shape[0] % B == 0:
        d = float((big[big.shape[0] // 2:] - sh).abs().max())
    else:
        continue
    if d > 0: print('%-28s %s differs by %.3e' % (name, tuple(big.shape), d))
