import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1)
for _ in range(3): bench.make_batch(0, 4, 40960, dev, gen, 'morton')
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): bench.make_batch(0, 4, 40960, dev, gen, 'morton')
torch.cuda.synchronize()
print('make_batch wall %.2f ms (includes host-side cloud synthesis + H2D)' % ((time.perf_counter() - t0) / 10 * 1e3))
import crfconv_amd, numpy as np
clouds = [bench.synth_cloud(i, 40960) for i in range(4)]
pos = torch.from_numpy(np.stack([c[0] for c in clouds])).to(dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): crfconv_amd.multiscale_compute(pos, generator=gen, sort='morton')
torch.cuda.synchronize()
print('multiscale_compute wall %.2f ms' % ((time.perf_counter() - t0) / 10 * 1e3))
