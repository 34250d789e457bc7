"""Which tensor shapes go through aten::copy_ / fill_ / add_ in one eager training step (torch.profiler, record_shapes)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, crfconv_amd
from crfconv_amd import models, ops, distributed as D
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
net = models.PointConvBig(6, 13, use_crf=True, steps=3).to(dev).train()
bucket = D.FlatGradAllReduce(net)
opt = crfconv_amd.optim.FlatSGD(bucket, lr=1e-2, momentum=0.95, weight_decay=1e-4)
cw = torch.ones(13, device=dev)
def part_a():
    for p in bucket.params: p.grad = None
    loss = ops.training_loss(net(data), data.y, cw, ignore_index=-1)
    with ops.deferred_weight_grads():
        loss.backward()
    torch._foreach_copy_(bucket.views, [p.grad for p in bucket.params])
for _ in range(3): part_a(); opt.step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    part_a(); opt.step(); torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if e.device_type.name != 'CPU' or not e.kernels: continue
    if e.name in ('aten::copy_', 'aten::fill_', 'aten::add_', 'aten::add', 'aten::cat', 'aten::mm', 'aten::addmm', 'aten::sum', 'aten::_foreach_copy_', 'aten::clone', 'aten::contiguous', 'aten::zero_', 'aten::zeros', 'aten::mul', 'aten::div_'):
        agg[(e.name, str(e.input_shapes)[:110], ','.join(sorted(set(k.name[:40] for k in e.kernels)))[:60])] += len(e.kernels)
for k, n in sorted(agg.items(), key=lambda kv: -kv[1])[:70]:
    print('%3d  %-22s %-112s %s' % (n, k[0], k[1], k[2]))
