"""Which host call sites still reach a vendor GEMM / framework elementwise kernel in one training step, with shapes."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, crfconv_amd
from crfconv_amd import models, ops
seen = collections.Counter()
def wrap(owner, name):
    orig = getattr(owner, name)
    def f(*a, **k):
        fr = [s for s in traceback.extract_stack()[:-1] if 'crfconv_amd' in s.filename]
        site = '%s:%d' % (os.path.basename(fr[-1].filename), fr[-1].lineno) if fr else '?'
        shapes = tuple(tuple(t.shape) for t in a if isinstance(t, torch.Tensor))
        seen[(name, site, shapes)] += 1
        return orig(*a, **k)
    setattr(owner, name, f)
for owner, name in ((torch.Tensor, '__matmul__'), (torch.Tensor, 'matmul'), (torch, 'addmm'), (torch.Tensor, 'addmm_'),
                    (torch.nn.functional, 'linear'), (torch, 'mm'), (torch, 'matmul'), (torch.Tensor, 'mm')):
    wrap(owner, name)
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
torch.manual_seed(0)
net = models.PointConvBig(6, 13, use_crf=True, steps=3).to(dev).train()
cw = torch.ones(13, device=dev)
for it in range(2):
    seen.clear()
    for p in net.parameters():
        p.grad = None
    loss = ops.training_loss(net(data), data.y, cw, ignore_index=-1)
    with ops.deferred_weight_grads():
        loss.backward()
torch.cuda.synchronize()
for (name, site, shapes), n in sorted(seen.items(), key=lambda kv: kv[0][1]):
    print('%-12s %-16s x%d  %s' % (name, site, n, shapes))
