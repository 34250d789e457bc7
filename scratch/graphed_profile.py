"""Where the host time of the reference loop on a train.GraphedModel goes (trainval.py:99-106 verbatim, model wrapped once).
usage: python3 scratch/graphed_profile.py > gpurun_out/graphed_profile.txt"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import torch.nn.functional as F
from crfconv_amd import models
from crfconv_amd.train import GraphedModel
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
torch.manual_seed(0)
net = GraphedModel(models.PointConvBig(6, 13, use_crf=True, steps=3).to(dev).train())
cw = torch.ones(13, device=dev)
opt = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.95, weight_decay=1e-4)


def one(sync=False, acc=None):
    def lap(name, t0):
        if sync:
            torch.cuda.synchronize()
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()
    t = time.perf_counter()
    opt.zero_grad()
    t = lap('zero_grad', t)
    out = net(data)
    t = lap('net(data)', t)
    loss = F.cross_entropy(out, data.y.reshape(-1) - 1, weight=cw, ignore_index=-1)
    t = lap('cross_entropy', t)
    loss.backward()
    t = lap('backward', t)
    opt.step()
    t = lap('opt.step', t)
    return loss.detach()


for _ in range(5):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    one()
torch.cuda.synchronize()
print('graphed-module ms per step: %.3f' % ((time.perf_counter() - t0) / 20 * 1e3))
acc = {}
for _ in range(20):
    one(True, acc)
print('synchronised laps (ms per step):', {k: round(v / 20 * 1e3, 3) for k, v in acc.items()})
t0 = time.perf_counter()
for _ in range(20):
    net.static.load_(data)
torch.cuda.synchronize()
print('static.load_(data) alone: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
for _ in range(20):
    net.fwd_graph.replay(); net.bwd_graph.replay()
torch.cuda.synchronize()
print('the two replays alone: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    one()
torch.cuda.synchronize()
pr.disable()
for key in ('tottime', 'cumulative'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(35)
    print(s.getvalue()[:7000])
