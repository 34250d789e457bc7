"""Coarse-level products C = A B^T with deep K (the dependent chunk hand-overs of gemm_kernel): graph replays, HIP events.
usage: [CRFCONV_LIB=scratch/variants/lib_<name>.so] python3 scratch/gemm_k_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crfconv_amd import ops
dev = 'cuda'
for M, N, K in ((640, 128, 512), (640, 512, 128), (2560, 256, 512), (2560, 128, 256), (2560, 256, 64), (10240, 128, 128), (10240, 256, 128), (10240, 32, 128)):
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) / 16
    def run():
        return ops._gemm(A, B, nk=True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): run()
        for _ in range(3): g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(20):
            e0.record(s); g.replay(); e1.record(s); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 100.0)
        ts.sort()
    print('%6d x %4d x %4d   %6.2f us per product' % (M, N, K, ts[len(ts) // 2]), flush=True)
