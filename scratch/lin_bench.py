"""linear_fwd_kernel (csrc/linear.hip) achieved bandwidth per shape of the fine levels: Y = X W^T with the BatchNorm
statistic records, graph-replayed so that launch overhead does not count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crfconv_amd import ops
dev = torch.device('cuda', 0)
shapes = [(163840, 32, 128), (163840, 128, 32), (163840, 32, 32), (163840, 8, 32), (163840, 32, 8), (163840, 64, 32),
          (40960, 64, 64), (40960, 16, 64), (40960, 64, 16), (40960, 128, 64), (10240, 128, 128), (10240, 32, 128)]
shapes = [sh for sh in shapes if sh[0] >= 40960]
for M, Ci, Co in shapes:
    x = torch.randn(M, Ci, device=dev)
    W = torch.randn(Co, Ci, device=dev) / Ci ** 0.5
    for stats in (True, False):
        ops._mfma_matmul(x, W, None, False, stats)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                ops._mfma_matmul(x, W, None, False, stats)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): g.replay()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 200 * 1e3
        print('m %6d  Ci %3d  Co %3d  stats %d: %6.1f us   %5.2f TB/s (X read + Y write)' % (M, Ci, Co, stats, us, 4 * M * (Ci + Co) / us / 1e6), flush=True)
