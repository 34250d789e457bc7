"""Config-2 whole-net eval: HIP vs oracle fp32 vs oracle fp64 (is the 1e-3 gap conditioning or a bug?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import _seeded as S
import crfconv_amd
from crfconv_amd import models
from oracle import crf_oracle as O
DEV = 'cuda:0'
seed_w = int(os.environ.get('SEEDW', 12)); T = int(os.environ.get('T', 3)); B = int(os.environ.get('B', 4)); N = int(os.environ.get('N', 40960))
g = torch.Generator().manual_seed(2)
rng = np.random.default_rng(2)
dims = np.array([200, 200, 75])
pos = np.empty((B, N, 3), np.float32)
for b in range(B):
    flat = rng.choice(int(dims.prod()), size=N, replace=False)
    ijk = np.stack(np.unravel_index(flat, dims), -1).astype(np.float64)
    pos[b] = ((ijk + 0.5) * 0.04 + rng.uniform(-0.01, 0.01, (N, 3))).astype(np.float32)
pos = torch.from_numpy(pos)
feats = torch.cat([pos, torch.rand(B, N, 3, generator=g)], -1)
data = crfconv_amd.multiscale_compute(pos.to(DEV), x=feats.to(DEV), generator=g)
net = models.PointConvBig(6, 13, use_crf=True, steps=T)
sd = S.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed_w)
net.load_state_dict(sd); net = net.to(DEV).eval()
with torch.no_grad():
    logits = net(data).cpu()
torch.set_num_threads(16)
ms = [{k: getattr(l, k).cpu() for k in ('pos', 'neighbor_idx', 'sub_idx', 'up_idx') if getattr(l, k, None) is not None} for l in data.multiscale]
with torch.no_grad():
    r32 = O.pointconv_resnet({k: v.clone() for k, v in sd.items()}, data.x.cpu(), ms, T, False, True)
    ms64 = [{k: (v.double() if v.is_floating_point() else v) for k, v in l.items()} for l in ms]
    r64 = O.pointconv_resnet({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, data.x.cpu().double(), ms64, T, False, True)
sc = max(1.0, float(r64.abs().max()))
print('max |ref64| %.2f' % sc)
print('HIP  vs fp64: %.3e (rel to max)   oracle32 vs fp64: %.3e   HIP vs oracle32: %.3e' % (
    float((logits.double() - r64).abs().max()) / sc, float((r32.double() - r64).abs().max()) / sc, float((logits - r32).abs().max()) / sc))
e = (logits.double() - r64).abs().max(1).values
print('rows with err > 1e-4*scale: HIP %d, oracle32 %d of %d' % (int((e > 1e-4 * sc).sum()), int(((r32.double() - r64).abs().max(1).values > 1e-4 * sc).sum()), e.numel()))
worst = int(e.argmax()); print('worst row', worst, 'cloud', worst // N, 'err', float(e[worst]))
