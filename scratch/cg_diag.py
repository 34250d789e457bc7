import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench, crfconv_amd
from crfconv_amd import models, ops, distributed as D
from crfconv_amd.data import CollateGraph
dev = torch.device('cuda', 0)
B, N = int(os.environ.get('B', 2)), int(os.environ.get('N', 4096))
gen = torch.Generator().manual_seed(1)
data, _ = bench.make_batch(0, B, N, dev, gen, 'morton')
net = models.PointConvBig(6, 13, True, 3).to(dev).train()
bucket = D.FlatGradAllReduce(net); opt = crfconv_amd.optim.FlatSGD(bucket, lr=1e-2, momentum=0.95, weight_decay=1e-4)
cw = torch.ones(13, device=dev)
def part_a():
    opt.zero_grad()
    loss = ops.training_loss(net(data), data.y, cw, ignore_index=-1)
    with ops.deferred_weight_grads():
        loss.backward()
    bucket.pack()
    return loss.detach()
mode = os.environ.get('MODE', 'graph')
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): part_a(); opt.step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
if os.environ.get('NOTRAIN'):
    pass
elif mode == 'graph':
    ga = torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga):
        sl = part_a()
    ga.replay(); torch.cuda.synchronize()
print('train graph ok', flush=True)
cg = CollateGraph(data, generator=torch.Generator().manual_seed(9))
raw = [bench.synth_cloud(70 + i, N) for i in range(B)]
pos = torch.from_numpy(np.stack([c[0] for c in raw])).to(dev)
x = torch.cat([pos, torch.from_numpy(np.stack([c[1] for c in raw])).to(dev)], -1)
y = torch.from_numpy(np.stack([c[2] for c in raw])).to(dev)
print('running collate graph', flush=True)
cg.run(pos, x, y); torch.cuda.synchronize(); print('captured + first replay ok', flush=True)
var = os.environ.get('VAR', '')
for i in range(3):
    if var == 'nodraw':
        cg.pos.copy_(pos); cg.graph.replay()
    elif var == 'replayonly':
        cg.graph.replay()
    elif var == 'eagerbetween':
        crfconv_amd.multiscale_compute(pos, x=x, y=y, choices=[c.clone() for c in cg.choices], sort='morton'); cg.run(pos, x, y)
    elif var == 'drawonly':
        cg._draw(); torch.cuda.synchronize(); cg.graph.replay()
    else:
        cg.run(pos, x, y)
    torch.cuda.synchronize(); print('replay', i, 'ok', flush=True)
if mode == 'graph':
    ga.replay(); torch.cuda.synchronize(); print('train replay after collate ok, loss', float(sl), flush=True)
import time
def tm(f, n=5):
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
from crfconv_amd.data import morton_order
print('input copies %.2f ms' % tm(lambda: (cg.pos.copy_(pos), cg.x.copy_(x), cg.y.copy_(y))))
print('_draw %.2f ms' % tm(cg._draw))
print('morton_order + copy %.2f ms' % tm(lambda: cg.order.copy_(morton_order(cg.pos))))
print('graph.replay %.2f ms' % tm(cg.graph.replay))
print('whole run %.2f ms' % tm(lambda: cg.run(pos, x, y)))
