"""Classifier head (MLP 32 -> 128, Dropout, Linear 128 -> 13) at 4 x 40 960 rows: stored form against the recomputing one
(csrc/head.hip), forward and forward + backward, HIP events around graph replays.  usage: python3 scratch/head_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crfconv_amd import ops
M, Ci, Co, C2 = 163840, 32, 128, 13
dev = 'cuda'
g = torch.Generator().manual_seed(1)
x0 = torch.randn(M, Ci, generator=g).to(dev)
W = torch.nn.Parameter((torch.randn(Co, Ci, generator=g) / 6).to(dev))
W2 = torch.nn.Parameter((torch.randn(C2, Co, generator=g) / 11).to(dev))
b2 = torch.nn.Parameter(torch.randn(C2, generator=g).to(dev))
go = torch.randn(M, C2, generator=g).to(dev)
bn = torch.nn.BatchNorm1d(Co).to(dev).train()
for rec in ((True,) if 'recompute' in sys.argv else (False, True)):
    for bwd in (False, True):
        x = x0.clone().requires_grad_(True)
        def run():
            out = ops.mlp_dropout_linear(x, W, bn, 0.1, 0.5, W2, b2, recompute=rec)
            if bwd:
                out.backward(go)
                x.grad = None; W.grad = None; W2.grad = None; b2.grad = None; bn.weight.grad = None; bn.bias.grad = None
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                run()
            for _ in range(5):
                gr.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(20):
                e0.record(s)
                for _ in range(10):
                    gr.replay()
                e1.record(s)
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 100.0)
            ts.sort()
        print('%-10s %-8s %7.1f us' % ('recompute' if rec else 'stored', 'fwd+bwd' if bwd else 'fwd', ts[len(ts) // 2]), flush=True)
