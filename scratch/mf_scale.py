import sys, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
dev = torch.device('cuda', 0)
for B in (1, 2, 4, 8, 16):
    gen = torch.Generator().manual_seed(1234)
    data, _ = bench.make_batch(0, B, 40960, dev, gen, 'morton')
    r = bench.roofline_meanfield(data, dev, 8, 3, iters=60)
    print('B=%2d m=%7d  avg %.1f us  min %.1f us  frac %.3f' % (B, B * 40960, r['avg_launch_us'], r['min_launch_us'], r['frac']))
