"""A/B timing of the step-kernel variants in scratch/mfv.hip on the bench's level-0 tables."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from crfconv_amd.graph import table_of, ptr, stream_ptr
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libmfv.so'))
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib.mfv_step.argtypes = [i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, i64, vp]
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
ms0 = data.multiscale[0]
B, N, K = ms0.neighbor_idx.shape
m, H = B * N, 8
tab = table_of(ms0.neighbor_idx, N)
idx16 = tab.idx16
# rows with columns 1..K-1 sorted by source id
loc = idx16.view(torch.int16).to(torch.int32) & 0xffff
srt = torch.cat([loc[:, :1], torch.sort(loc[:, 1:], dim=1).values], 1).to(torch.int16).contiguous()
g = torch.Generator().manual_seed(1)
z = torch.randn(m, H, generator=g).to(dev); xin = torch.randn(m, H, generator=g).to(dev)
s = torch.rand(m, K, generator=g).to(dev); s[:, 0] = 0; s /= s.sum(1, keepdim=True)
c = torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)
C = c.t() @ c; Q = torch.linalg.inv(torch.eye(H) + C); P = (C @ Q).to(dev).contiguous(); Q = Q.to(dev).contiguous()
st = stream_ptr()
names = {0: 'base', 1: 'own-row (no gather)', 2: 'nontemporal streams', 3: 'two gather batches', 4: 'win 512/128',
         5: 'win 1024/256', 6: 'win 256/64', 7: 'pipe x2', 8: 'pipe x4', 9: 'win 1024/512', 10: 'coalesced rows via LDS', 11: 'coalesced rows + 2 batches', 12: '2 batches, nt store', 13: '2 batches, nt idx/s/z loads, nt store'}
win = (loc[:, 1:] - torch.arange(m, device=dev).remainder(N)[:, None]).abs()
for h in (64, 128, 256, 512, 1024):
    print('neighbour refs within +-%d rows: %.3f' % (h, float((win <= h).float().mean())))
ref = None
for label, table in (('knn-order', idx16), ('sorted', srt)):
    for v in ([int(a) for a in os.environ.get('STEPV', '').split(',') if a] or sorted(names)):
        out = torch.empty(m, H, device=dev)
        def launch():
            rc = lib.mfv_step(v, ptr(xin), ptr(z), ptr(s), ptr(table), N, N, ptr(Q), ptr(P), ptr(out), m, st)
            assert rc == 0, rc
        for _ in range(5): launch()
        torch.cuda.synchronize()
        n = 200
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): launch()
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / n * 1e3
        if v == 0 and label == 'knn-order': ref = out.clone()
        err = float((out - ref).abs().max()) if (v != 1 and label == 'knn-order') else float('nan')
        print('%-10s v%d %-22s %7.2f us/launch (back-to-back)  err %.1e' % (label, v, names[v], t, err), flush=True)

# ---- first kernel (similarity + step 1)
lib.mfv_sim.argtypes = [i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, i64, vp]
y = torch.randn(m, H, generator=g).to(dev)
snames = {0: 'base occ4', 1: 'own-row', 2: 'packed idx, 8-batches, minw5', 3: 'y/z interleaved 8-batches', 4: 'packed 8-batches minw6',
          5: 'y/z interleaved minw5', 6: 'base minw5', 7: 'own-row, no s store', 8: 'own-row, nt s store', 9: 'interleaved, nt s store', 10: 'interleaved, nt s+x store', 11: 'interleaved, no s store', 12: 'interleaved, s via LDS', 13: 'base, s via LDS', 14: 'own-row, s via LDS', 15: 'base, s via LDS nt', 16: 'base, s via LDS nt, x1 nt'}
sv = [int(a) for a in os.environ.get('SIMV', '').split(',') if a] or sorted(snames)
ref_s = ref_x = None
for label, table in (('knn-order', idx16), ('sorted', srt)):
    for v in sv:
        so = torch.empty(m, K, device=dev); xo = torch.empty(m, H, device=dev)
        def launch():
            rc = lib.mfv_sim(v, ptr(y), ptr(z), ptr(table), N, N, ptr(Q), ptr(P), ptr(so), ptr(xo), m, st)
            assert rc == 0, rc
        for _ in range(5): launch()
        torch.cuda.synchronize()
        n = 200
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): launch()
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / n * 1e3
        if v == 0 and label == 'knn-order': ref_s, ref_x = so.clone(), xo.clone()
        ok = v != 1 and label == 'knn-order' and ref_s is not None
        print('sim %-10s v%d %-30s %7.2f us/launch  err s %.1e x %.1e' % (label, v, snames.get(v, '?'), t,
              float((so - ref_s).abs().max()) if ok else float('nan'), float((xo - ref_x).abs().max()) if ok else float('nan')), flush=True)

# ---- whole forward in sequence: sim variant, then two step variants
xs = torch.empty(3, m, H, device=dev); so = torch.empty(m, K, device=dev)
for sv_, tv_ in ((0, 0), (13, 3), (15, 3), (16, 3), (13, 12), (15, 12), (16, 12), (16, 13), (13, 13)):
    def launch():
        lib.mfv_sim(sv_, ptr(y), ptr(z), ptr(srt), N, N, ptr(Q), ptr(P), ptr(so), ptr(xs[0]), m, st)
        lib.mfv_step(tv_, ptr(xs[0]), ptr(z), ptr(so), ptr(srt), N, N, ptr(Q), ptr(P), ptr(xs[1]), m, st)
        lib.mfv_step(tv_, ptr(xs[1]), ptr(z), ptr(so), ptr(srt), N, N, ptr(Q), ptr(P), ptr(xs[2]), m, st)
    for _ in range(5): launch()
    torch.cuda.synchronize()
    n = 200
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): launch()
    b.record(); torch.cuda.synchronize()
    print('seq sim v%d + 2 x step v%d: %7.2f us per forward (sorted table)' % (sv_, tv_, a.elapsed_time(b) / n * 1e3), flush=True)

# ---- interleaved yz table, four lanes per point
lib.mfv_sim_yz.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, i64, vp]
yz = torch.cat([y, z], 1).contiguous()
so2 = torch.empty(m, K, device=dev); xo2 = torch.empty(m, H, device=dev)
lib.mfv_sim(13, ptr(y), ptr(z), ptr(srt), N, N, ptr(Q), ptr(P), ptr(so), ptr(xs[0]), m, st)
lib.mfv_sim_yz(ptr(yz), ptr(srt), N, N, ptr(Q), ptr(P), ptr(so2), ptr(xo2), m, st)
torch.cuda.synchronize()
print('yz variant err s %.1e x %.1e' % (float((so2 - so).abs().max()), float((xo2 - xs[0]).abs().max())))
for name, fn in (('sim v13 (separate y, z)', lambda: lib.mfv_sim(13, ptr(y), ptr(z), ptr(srt), N, N, ptr(Q), ptr(P), ptr(so), ptr(xs[0]), m, st)),
                 ('sim yz interleaved', lambda: lib.mfv_sim_yz(ptr(yz), ptr(srt), N, N, ptr(Q), ptr(P), ptr(so2), ptr(xo2), m, st))):
    for seq in (False, True):
        def launch():
            fn()
            if seq:
                lib.mfv_step(3, ptr(xs[0]), ptr(z), ptr(so), ptr(srt), N, N, ptr(Q), ptr(P), ptr(xs[1]), m, st)
                lib.mfv_step(3, ptr(xs[1]), ptr(z), ptr(so), ptr(srt), N, N, ptr(Q), ptr(P), ptr(xs[2]), m, st)
        for _ in range(5): launch()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200): launch()
        b.record(); torch.cuda.synchronize()
        print('%-26s %s: %7.2f us' % (name, 'whole forward (+2 steps)' if seq else 'alone, back-to-back', a.elapsed_time(b) / 200 * 1e3), flush=True)

# (measured and dropped: clouds 0-1 / 2-3 on two streams with an event fork/join per forward -- 62 us per forward against
#  25 us in one stream; the cross-queue dependencies cost far more than the launch boundaries they were meant to hide)
