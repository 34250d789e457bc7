"""Collate + table refresh of one batch, 10 times (for rocprofv3 --kernel-trace): where do the 5 ms per batch go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench, crfconv_amd
from crfconv_amd import models, ops
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
net = models.PointConvBig(6, 13, True, 3).to(dev).train()
loss = ops.training_loss(net(data), data.y, None, ignore_index=-1); loss.backward()       # builds every table / reverse CSR / moments
clouds = [bench.synth_cloud(50 + i, 40960) for i in range(4)]
pos = torch.from_numpy(np.stack([c[0] for c in clouds])).to(dev)
x = torch.cat([pos, torch.from_numpy(np.stack([c[1] for c in clouds])).to(dev)], -1)
y = torch.from_numpy(np.stack([c[2] for c in clouds])).to(dev)
torch.cuda.synchronize()
tc, tl = [], []
for it in range(10):
    t0 = time.perf_counter()
    nd = crfconv_amd.multiscale_compute(pos, x=x, y=y, generator=gen, sort='morton')
    torch.cuda.synchronize(); t1 = time.perf_counter()
    data.load_(nd)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    tc.append(t1 - t0); tl.append(t2 - t1)
print('collate median %.2f ms, load_ median %.2f ms' % (np.median(tc) * 1e3, np.median(tl) * 1e3))
