"""Cost of running K independent captured graphs on K streams (event fork / join around them) against one graph that holds all
of them in sequence: is the end-of-pass block of the training step (independent job families, ~0.37 ms serial) worth spreading over
streams?  Families = chains of small matrix products on few workgroups.  usage: python3 scratch/streams_fork_test.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crfconv_amd
from crfconv_amd import ops
dev = 'cuda'
K, L = 4, 5
A = [torch.randn(2560, 256, device=dev) for _ in range(K)]
W = [torch.randn(256, 256, device=dev) / 16 for _ in range(K)]
out = [None] * K
pre = torch.randn(163840, 32, device=dev)
def family(k):
    x = A[k]
    for _ in range(L):
        x = ops._gemm(x, W[k])
    out[k] = x
def head_like():
    return pre * 2.0
main = torch.cuda.Stream()
sides = [torch.cuda.Stream() for _ in range(K)]
with torch.cuda.stream(main):
    for k in range(K):
        family(k)
    head_like()
    torch.cuda.synchronize()
    g_all = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_all, stream=main):
        head_like()
        for k in range(K):
            family(k)
        head_like()
    g_pre, g_post = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_pre, stream=main):
        head_like()
    with torch.cuda.graph(g_post, stream=main):
        head_like()
gk = []
for k in range(K):
    with torch.cuda.stream(sides[k]):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=sides[k]):
            family(k)
        gk.append(g)
torch.cuda.synchronize()
ev0 = torch.cuda.Event()
evk = [torch.cuda.Event() for _ in range(K)]
def serial():
    with torch.cuda.stream(main):
        g_all.replay()
def forked(nstreams):
    with torch.cuda.stream(main):
        g_pre.replay()
        ev0.record(main)
    for k in range(K):
        s = sides[k % nstreams] if nstreams > 0 else main
        with torch.cuda.stream(s):
            if nstreams > 0:
                s.wait_event(ev0)
            gk[k].replay()
            if nstreams > 0:
                evk[k].record(s)
    with torch.cuda.stream(main):
        if nstreams > 0:
            for k in range(K):
                main.wait_event(evk[k])
        g_post.replay()
def timeit(fn, n=20):
    """GPU time of one call: a long spin kernel lets the host queue everything first (the real step has 4 ms of host slack)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        with torch.cuda.stream(main):
            torch.cuda._sleep(20_000_000)          # ~8 ms
            e0.record(main)
        for _ in range(5):
            fn()
        with torch.cuda.stream(main):
            e1.record(main)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 5)
    ts.sort()
    return ts[len(ts) // 2]
print('one graph, one stream            %8.1f us' % timeit(serial))
print('separate graphs, one stream      %8.1f us' % timeit(lambda: forked(0)))
for ns in (1, 2, 4):
    print('separate graphs on %d side streams %7.1f us' % (ns, timeit(lambda: forked(ns))))
