"""Coarse-level Linear -> BatchNorm -> LeakyReLU: the one-launch kernel (csrc/mlp_small.hip) against vendor GEMM + bn_small,
forward and forward+backward, per shape of PointConvBig's levels 3-5 at config 2.  python3 scratch/mlps_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crfconv_amd import ops

dev = torch.device('cuda', 0)
shapes = [(2560, 128, 64), (2560, 64, 256), (2560, 128, 256), (2560, 256, 64), (2560, 256, 256), (1280, 256, 128),
          (1280, 128, 512), (1280, 256, 512), (1280, 512, 128), (1280, 512, 512), (2560, 512, 256), (2560, 512, 256)]


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return float(np.median(ts))


for M, Ci, Co in shapes:
    x = torch.randn(M, Ci, device=dev, requires_grad=True)
    W = (torch.randn(Co, Ci, device=dev) / Ci ** 0.5).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(Co).to(dev).train()
    go = torch.randn(M, Co, device=dev)
    res = {}
    for small in (True, False):
        ops._NO_SMALL_MLP_ENV = not small

        def fwd():
            with torch.no_grad():
                if small:
                    return ops.mlp_block(x, W, bn, 0.1)
                y, rec = ops.linear(x, W, None, want_stats=True)
                return ops.bn_act(y, bn, True, 0.1, records=rec)

        def fb():
            if small:
                out = ops.mlp_block(x, W, bn, 0.1)
            else:
                y, rec = ops.linear(x, W, None, want_stats=True)
                out = ops.bn_act(y, bn, True, 0.1, records=rec)
            out.backward(go)
        g = torch.cuda.CUDAGraph()
        fwd(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(10):
                fwd()
        res[small] = (timeit(g.replay, 20) / 10, timeit(fb, 20))
    print('m %5d  Ci %3d  Co %3d   forward (graph replay): one launch %6.1f us   vendor GEMM + bn_small %6.1f us     fwd+bwd eager: %6.1f / %6.1f us' % (
        M, Ci, Co, res[True][0], res[False][0], res[True][1], res[False][1]), flush=True)
