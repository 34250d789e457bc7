"""Where a loop over crops blocks the host: host time of each call of the tiled-scene pipeline WITHOUT synchronising, then one synchronisation.
usage: python3 scratch/sync_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crfconv_amd
from crfconv_amd import models
from crfconv_amd.data import multiscale_compute
from crfconv_amd.sampling import PossibilitySampler, VoteAccumulator
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(50)
n_scene, n_crop, K, T, C = 1 << 20, 65536, 32, 5, 8
pts = (torch.rand(n_scene, 3, generator=g) * torch.tensor([60.0, 60.0, 15.0])).to(dev)
rgb = torch.rand(n_scene, 3, generator=g).to(dev)
net = models.PointConvBig(6, C, use_crf=True, steps=T).to(dev).eval()
smp = PossibilitySampler([pts], rgb=[rgb], num_points=n_crop, split='test', generator=torch.Generator().manual_seed(51))
votes = VoteAccumulator([n_scene], C, device=dev)
gen = torch.Generator().manual_seed(52)
def timed(name, fn, acc):
    t0 = time.perf_counter(); out = fn(); acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3); return out
for rnd in range(2):
    acc = {}
    torch.cuda.synchronize(); t_all = time.perf_counter()
    for i in range(8):
        crop = timed('get_random', smp.get_random, acc)
        pos = crop.pos.unsqueeze(0); x = torch.cat([crop.pos, crop.rgb], -1).unsqueeze(0)
        data = timed('multiscale_compute', lambda: multiscale_compute(pos, x=x, point_idx=crop.point_idx.unsqueeze(0), kernel_size=(K,) * 5, generator=gen, sort='morton'), acc)
        with torch.no_grad():
            logits = timed('net', lambda: net(data), acc)
        timed('votes.update', lambda: votes.update(data.point_idx, crop.cloud, logits=logits), acc)
    t_host = (time.perf_counter() - t_all) * 1e3
    t0 = time.perf_counter(); torch.cuda.synchronize(); t_sync = (time.perf_counter() - t0) * 1e3
    print('round %d: 8 crops, host %.1f ms + final synchronise %.1f ms' % (rnd, t_host, t_sync))
    for k, v in acc.items():
        print('   %-20s host ms per call: %s' % (k, ' '.join('%.2f' % x for x in v)))
# the sync itself
for n in (1, 1, 1):
    a = torch.zeros(16, device=dev); a += 1
    t0 = time.perf_counter(); torch.cuda.synchronize(); print('synchronize behind one tiny kernel: %.3f ms' % ((time.perf_counter() - t0) * 1e3))
    t0 = time.perf_counter(); v = a[0].item(); print('.item(): %.3f ms' % ((time.perf_counter() - t0) * 1e3))
ev = torch.cuda.Event(); a += 1; ev.record()
t0 = time.perf_counter(); ev.synchronize(); print('event.synchronize: %.3f ms' % ((time.perf_counter() - t0) * 1e3))
a += 1; ev.record(); t0 = time.perf_counter()
while not ev.query(): pass
print('event.query spin: %.3f ms' % ((time.perf_counter() - t0) * 1e3))
print('env', {k: v for k, v in os.environ.items() if k.startswith(('HIP', 'HSA', 'GPU_', 'AMD', 'ROC'))})
