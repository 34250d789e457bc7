"""Which Python lines issue the small framework kernels of one training step (torch.profiler, eager)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, crfconv_amd
from crfconv_amd import models, ops
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
net = models.PointConvBig(6, 13, use_crf=True, steps=3).to(dev).train()
opt = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.95, weight_decay=1e-4)
cw = torch.ones(13, device=dev)
def step():
    for p in net.parameters(): p.grad = None
    loss = ops.training_loss(net(data), data.y, cw, ignore_index=-1)
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
evs = prof.events()
agg = collections.Counter(); tim = collections.Counter()
for e in evs:
    if e.device_type.name != 'CPU' or not e.kernels: continue
    if not e.name.startswith('aten::'): continue
    frame = next((s for s in (e.stack or []) if 'crfconv_amd' in s or 'bench.py' in s or 'scratch' in s), 'autograd/other')
    key = (e.name, frame.split('/root/repo/')[-1][:90] if 'repo' in frame else frame[:90])
    agg[key] += len(e.kernels); tim[key] += sum(k.duration for k in e.kernels)
for k, n in agg.most_common(60):
    print('%4d launches %7.1f us  %-28s %s' % (n, tim[k], k[0], k[1]))
