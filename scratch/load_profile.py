"""Where the host time of MultiScaleData.load_ goes (22 ms per fresh batch eagerly against 0.9 ms as a graph).
usage: python3 scratch/load_profile.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import crfconv_amd
from crfconv_amd import models
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
net = models.PointConvBig(6, 13, use_crf=True, steps=3).to(dev).train()
out = net(data); out.sum().backward()            # builds every table / reverse CSR / moments memo the step uses
others = [bench.make_batch(1 + i, 4, 40960, dev, gen, 'morton')[0] for i in range(3)]
torch.cuda.synchronize()
for defer in (False, True):
    for nd in others:
        data.load_(nd, defer_check=defer)
    torch.cuda.synchronize()
    ts = []
    for i in range(12):
        t0 = time.perf_counter()
        data.load_(others[i % 3], defer_check=defer)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((t1 - t0, time.perf_counter() - t0))
    ts.sort()
    print('defer_check=%s: host %.2f ms  host+device %.2f ms (medians of 12)' % (defer, ts[6][0] * 1e3, sorted(t[1] for t in ts)[6] * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(12):
    data.load_(others[i % 3], defer_check=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
