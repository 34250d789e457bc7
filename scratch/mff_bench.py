"""Fused (one-launch) vs per-step mean-field forward on the bench's level-0 tables: bit parity, a staleness check with
inputs that change between launches, and back-to-back timing.  Run on the GPU box: python3 scratch/mff_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from crfconv_amd import _lib
from crfconv_amd.graph import table_of, ptr, stream_ptr

dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
B, N = int(os.environ.get('B', 4)), int(os.environ.get('N', 40960))
H, T = int(os.environ.get('H', 8)), int(os.environ.get('T', 3))
data, _ = bench.make_batch(0, B, N, dev, gen, 'morton')
ms0 = data.multiscale[0]
K = ms0.neighbor_idx.shape[2]
m = B * N
tab = table_of(ms0.neighbor_idx, N)
g = torch.Generator().manual_seed(1)
z = torch.randn(m, H, generator=g).to(dev)
y = torch.randn(m, H, generator=g).to(dev)
c = torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)
C = c.t() @ c
Q = torch.linalg.inv(torch.eye(H) + C)
P = (C @ Q).to(dev).contiguous()
Q = Q.to(dev).contiguous()
st = stream_ptr()
lib = _lib.load()
wsb = lib.crfconv_meanfield_fused_workspace()
ws = torch.zeros(wsb // 4, dtype=torch.int32, device=dev)
print('fused supported:', lib.crfconv_meanfield_fused_supported(m, H, K, 1, T), 'm', m, 'H', H, 'K', K, 'T', T, flush=True)

s_ref = torch.empty(m, K, device=dev); xs_ref = torch.empty(T, m, H, device=dev)
s_f = torch.empty(m, K, device=dev); xs_f = torch.empty(T, m, H, device=dev)


def unfused(zz, yy, s, xs):
    _lib.call('crfconv_meanfield_forward_u16', ptr(zz), ptr(yy), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src,
              K, 1, m, H, ptr(Q), ptr(P), T, ptr(s), ptr(xs), st)


def fused(zz, yy, s, xs):
    _lib.call('crfconv_meanfield_forward_fused', ptr(zz), ptr(yy), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src,
              K, 1, m, H, ptr(Q), ptr(P), T, ptr(s), ptr(xs), ptr(ws), wsb, st)


unfused(z, y, s_ref, xs_ref)
xs_f.fill_(float('nan')); s_f.fill_(float('nan'))
fused(z, y, s_f, xs_f)
torch.cuda.synchronize()
print('give-up word:', int(ws[17 * 32]))
print('max |s - ref| %.3e   max |xs - ref| per step %s   bit-equal %s' % (
    float((s_f - s_ref).abs().max()), [float((xs_f[t] - xs_ref[t]).abs().max()) for t in range(T)],
    bool(torch.equal(xs_f, xs_ref) and torch.equal(s_f, s_ref))), flush=True)

# staleness: the SAME output buffers, inputs changing every launch, fused launches back to back (no sync between)
bad = 0
zs = [torch.randn(m, H, generator=g).to(dev) for _ in range(6)]
refs = []
for zz in zs:
    unfused(zz, y, s_ref, xs_ref)
    refs.append(xs_ref.clone())
outs = []
for rep in range(4):
    for zz in zs:
        fused(zz, y, s_f, xs_f)
        outs.append(xs_f.clone())
torch.cuda.synchronize()
for i, o in enumerate(outs):
    if not torch.equal(o, refs[i % len(zs)]):
        bad += 1
print('staleness check: %d of %d launches differ from the per-step kernels' % (bad, len(outs)), flush=True)


def timeit(fn, n=200, reps=5):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return min(ts), float(np.median(ts))


alg = m * (4 * (K - 1) + 4 * H * (2 * T + 1))
for name, fn in (('per-step launches (s stored)', lambda: unfused(z, y, s_ref, xs_ref)),
                 ('fused, s stored', lambda: fused(z, y, s_f, xs_f)),
                 ('fused, no s', lambda: fused(z, y, None, xs_f))):
    lo, med = timeit(fn)
    print('%-30s min %7.2f us  median %7.2f us   frac of 8 TB/s at median: %.3f' % (name, lo, med, alg / (med * 1e-6) / 8e12), flush=True)
print('give-up word:', int(ws[17 * 32]))

# ---- where the time goes: the fused launch at T = 1, 2, 3, 5 (phase 0 alone, then +1 barrier + step each), and the memset
for TT in (1, 3):
    xs_t = torch.empty(TT, m, H, device=dev)
    def f_t():
        _lib.call('crfconv_meanfield_forward_fused', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src,
                  K, 1, m, H, ptr(Q), ptr(P), TT, None, ptr(xs_t), ptr(ws), wsb, st)
    def u_t():
        _lib.call('crfconv_meanfield_forward_u16', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src,
                  K, 1, m, H, ptr(Q), ptr(P), TT, ptr(s_ref), ptr(xs_t), st)
    print('T=%d  fused (no s) %7.2f us   per-step launches %7.2f us' % (TT, timeit(f_t)[1], timeit(u_t)[1]), flush=True)
print('memset of the sync words alone: %.2f us' % timeit(lambda: ws.zero_())[1])

# ---- in-kernel stamps (diagnostic build): per-workgroup phase times, 100 MHz ticks
import ctypes
lib.crfconv_meanfield_forward_fused_stamps.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 5
if H == 8 and K == 16:
    grid = (m + 639) // 640
    dbg = torch.zeros(grid, 64, dtype=torch.int64, device=dev)
    for rep in range(3):
        rc = lib.crfconv_meanfield_forward_fused_stamps(ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src, m,
                                                        ptr(Q), ptr(P), T, None, ptr(xs_f), ptr(ws), ptr(dbg), st)
        assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)
    t0 = d[:, 0].min()
    us = lambda a: (a - t0) / 100.0
    def line(name, col):
        v = us(d[:, col])
        print('  %-28s min %6.2f  median %6.2f  max %6.2f us after the first workgroup started' % (name, v.min(), np.median(v), v.max()))
    line('workgroup start', 0)
    line('phase-0 result ready', 1)
    for t in range(1, T):
        line('step %d: own stores drained' % t, 8 * t)
        line('step %d: left the barrier' % t, 8 * t + 1)
        line('step %d: gathers consumed' % t, 8 * t + 2)
        line('step %d: x_t stored (issued)' % t, 8 * t + 3)
    line('end (stores drained)', 7)
    for t in range(1, T):
        rel = d[:, 8 * t + 6]
        rel = rel[rel > 0]
        print('  barrier %d: workgroups arrived %.2f .. %.2f us, own atomic back (median) +%.2f us, release issued at %s us, left %.2f .. %.2f us' % (
            t, us(d[:, 8 * t + 4]).min(), us(d[:, 8 * t + 4]).max(), np.median(d[:, 8 * t + 5] - d[:, 8 * t + 4]) / 100,
            ['%.2f' % v for v in us(rel)], us(d[:, 8 * t + 1]).min(), us(d[:, 8 * t + 1]).max()))
    for t in range(1, T):
        print('  step %d medians: barrier wait %.2f us, gather %.2f us, matvec+store issue %.2f us' % (
            t, np.median(d[:, 8 * t + 1] - d[:, 8 * t]) / 100, np.median(d[:, 8 * t + 2] - d[:, 8 * t + 1]) / 100,
            np.median(d[:, 8 * t + 3] - d[:, 8 * t + 2]) / 100))
