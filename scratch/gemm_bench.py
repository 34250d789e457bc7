"""crfconv_gemm against the vendor GEMM on the shapes of the training step: error vs float64 and time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crfconv_amd
from crfconv_amd import _lib
from crfconv_amd.ops import ptr, stream_ptr
dev = torch.device('cuda', 0)
SH = [(640, 128, 512, 0), (2560, 64, 256, 0), (2560, 512, 256, 0), (2560, 64, 64, 0), (2560, 256, 64, 0), (640, 64, 64, 0),
      (640, 512, 64, 0), (640, 512, 128, 0), (2560, 256, 512, 0), (2560, 256, 128, 0), (10240, 128, 128, 0), (40960, 64, 64, 0),
      (163840, 32, 32, 0), (2560, 32, 256, 1), (2560, 32, 32, 1), (10240, 128, 256, 1), (2560, 256, 32, 0), (10240, 256, 128, 0),
      (1000, 36, 20, 0), (77, 12, 8, 1)]
def timeit(fn, n=50):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): fn()
    gr.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4): gr.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (4 * n) * 1e3
g = torch.Generator(device='cpu').manual_seed(0)
tot_o = tot_v = 0.0
for M, N, K, nk in SH:
    A = torch.randn(M, K, generator=g).to(dev)
    B = (torch.randn(N, K, generator=g) if nk else torch.randn(K, N, generator=g)).to(dev)
    add = torch.randn(M, N, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.empty(M, N, device=dev)
    ref = A.double() @ (B.double().t() if nk else B.double()) + bias.double() + add.double()
    _lib.call('crfconv_gemm', ptr(A), ptr(B), ptr(bias), ptr(add), M, N, K, nk, ptr(C), stream_ptr())
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    vend = (A @ (B.t() if nk else B)) + bias + add
    verr = float((vend.double() - ref).abs().max() / ref.abs().max())
    t_own = timeit(lambda: _lib.call('crfconv_gemm', ptr(A), ptr(B), None, None, M, N, K, nk, ptr(C), stream_ptr()))
    t_ven = timeit(lambda: torch.mm(A, B.t() if nk else B, out=C))
    tot_o += t_own; tot_v += t_ven
    print('%7d x %3d x %3d %s  err %.1e (vendor %.1e)  own %6.1f us  vendor %6.1f us' % (M, N, K, 'NT' if nk else 'NN', err, verr, t_own, t_ven), flush=True)
print('sum own %.1f us  vendor %.1f us' % (tot_o, tot_v))
