"""Framework (aten) ops that still launch device kernels inside the benchmarked step, with the Python site of each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, crfconv_amd
from crfconv_amd import models, ops, distributed as D
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
torch.manual_seed(0)
net = models.PointConvBig(6, 13, use_crf=True, steps=3).to(dev).train()
bucket = D.FlatGradAllReduce(net)
opt = crfconv_amd.optim.FlatSGD(bucket, lr=1e-2, momentum=0.95, weight_decay=1e-4)
cw = torch.ones(13, device=dev)
unit = torch.ones((), device=dev)
def step():
    opt.zero_grad()
    loss = ops.training_loss(net(data), data.y, cw, ignore_index=-1)
    with ops.deferred_weight_grads(sink=bucket.view_of):
        loss.backward(unit)
    bucket.pack()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
import traceback, collections
from torch.utils._python_dispatch import TorchDispatchMode
seen = collections.Counter()
SKIP = ('aten::empty', 'aten::view', 'aten::_unsafe_view', 'aten::detach', 'aten::as_strided', 'aten::t', 'aten::transpose', 'aten::reshape',
        'aten::slice', 'aten::select', 'aten::alias', 'aten::expand', 'aten::unsqueeze', 'aten::squeeze', 'aten::permute', 'aten::narrow',
        'aten::split', 'aten::unbind', 'aten::_local_scalar_dense', 'aten::lift_fresh', 'aten::new_empty', 'aten::empty_like', 'aten::empty_strided')
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name()
        if not any(name.startswith(k) for k in SKIP):
            fr = [f for f in traceback.extract_stack()[:-1] if 'crfconv_amd/' in f.filename]
            site = ' < '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in fr[-3:][::-1]) if fr else '(engine)'
            shapes = tuple(tuple(a.shape) if isinstance(a, torch.Tensor) else (len(a) if isinstance(a, (list, tuple)) else a) for a in args)
            seen[(name, site, str(shapes)[:100])] += 1
        return func(*args, **(kwargs or {}))
with Log():
    step()
torch.cuda.synchronize()
for (name, site, shapes), n in sorted(seen.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    print('%-26s x%-3d %s   %s' % (name, n, site, shapes))
