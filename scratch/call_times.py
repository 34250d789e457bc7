"""Device time of every library call of one eager training step, grouped by entry point and integer arguments (shapes):
events around each call (the host is slower than the device in eager mode, so an interval = that call's kernels)."""
import os, sys, collections, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, crfconv_amd
from crfconv_amd import models, ops, _lib, distributed as D
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
orig = _lib.call
log = []
def timed(name, *args):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = orig(name, *args)
    b.record()
    ints = tuple(x for x in args if isinstance(x, int) and not isinstance(x, bool) and abs(x) < (1 << 24))
    log.append((name, ints, a, b))
    return r
reps = 5
_lib.call = timed
for _ in range(reps):
    step()
torch.cuda.synchronize()
_lib.call = orig
agg = collections.OrderedDict()
for name, ints, a, b in log:
    k = (name, ints)
    t = agg.setdefault(k, [0, 0.0])
    t[0] += 1
    t[1] += a.elapsed_time(b) * 1e3
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values()) / reps
print('library calls per step: %d, summed device time %.1f us' % (len(log) // reps, tot))
byname = collections.Counter()
for (name, ints), (n, t) in agg.items():
    byname[name] += t / reps
for name, t in byname.most_common():
    print('  %-44s %8.1f us' % (name, t))
print()
for (name, ints), (n, t) in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 120]:
    print('%-40s x%-3d %8.1f us/step %7.1f avg  %s' % (name.replace('crfconv_', ''), n // reps, t / reps, t / n, ints))
