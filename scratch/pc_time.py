"""PointConv forward / forward+backward graph-replay times at level 0 (d = 8) and level 1 (d = 16) of the bench batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
r = bench.roofline_pointconv(data, dev, 8)
print('level 0 d=8 : fwd %.1f us  fwd+bwd %.1f us' % (r['avg_launch_us'], r['fwd_bwd_us']), flush=True)
class L1: pass
d1 = L1(); d1.multiscale = data.multiscale[1:]
r = bench.roofline_pointconv(d1, dev, 16)
print('level 1 d=16: fwd %.1f us  fwd+bwd %.1f us' % (r['avg_launch_us'], r['fwd_bwd_us']), flush=True)
