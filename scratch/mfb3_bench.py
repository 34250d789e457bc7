"""Level-0 (or H / N of the environment) mean-field backward (csrc/crf_bwd.hip, T + 1 launches) against a float64 torch
reference on the device, bitwise reproducibility, HIP-event time.
env: B, N, H, T, KNN (neighbours), CRFCONV_LIB (A/B build of the library), HUB=1 (every row also points at row 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from crfconv_amd import _lib, ops
from crfconv_amd.graph import table_of, ptr, stream_ptr

dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
B, N = int(os.environ.get('B', 4)), int(os.environ.get('N', 40960))
H, T = int(os.environ.get('H', 8)), int(os.environ.get('T', 3))
KNN = int(os.environ.get('KNN', 16))
from crfconv_amd.data import morton_order
from crfconv_amd.utils import nearest_neighbors
pos = torch.from_numpy(np.stack([bench.synth_cloud(i, N)[0] for i in range(B)])).to(dev)
order = morton_order(pos)
pos = torch.gather(pos, 1, order[:, :, None].expand(-1, -1, 3)).contiguous()
idx = nearest_neighbors.knn_batch_device(pos, pos, KNN).clone()
if os.environ.get('HUB'):
    idx[:, :, 1] = 0                       # a hub: in-degree N
K = idx.shape[2]
m = B * N
tab = table_of(idx, N)
rev_ptr, rev_eid = tab.reverse
g = torch.Generator().manual_seed(1)
z = torch.randn(m, H, generator=g).to(dev)
y = (0.5 * torch.randn(m, H, generator=g)).to(dev)
c = torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)
C = c.t() @ c
Q = torch.linalg.inv(torch.eye(H) + C).to(dev).contiguous()
P = (C @ torch.linalg.inv(torch.eye(H) + C)).to(dev).contiguous()
gout = torch.randn(m, H, generator=g).to(dev)
lib = _lib.load()
st = stream_ptr()
s = torch.empty(m, K, device=dev)
xs = torch.empty(T, m, H, device=dev)
_lib.call('crfconv_meanfield_forward_u16', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src,
          K, 1, m, H, ptr(Q), ptr(P), T, ptr(s), ptr(xs), st)
inside = lib.crfconv_meanfield_backward_param_grads_inside(H) == 1


def buffers():
    d = dict(Gs=torch.empty(T, m, H, device=dev), aux=torch.empty(T, m, H, device=dev),
             dz=torch.empty(m, H, device=dev), dy_self=torch.empty(m, H, device=dev), dy=torch.empty(m, H, device=dev),
             w=torch.empty(m, K, device=dev), dP=torch.empty(H, H, device=dev), dQ=torch.empty(H, H, device=dev),
             mts=None if inside else torch.empty(T, m, H, device=dev), sumG=None if inside else torch.empty(m, H, device=dev))
    return d


def make(name, wsname):
    bufs = buffers()
    wsb = getattr(lib, wsname)(m, H, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ticket = ops._ticket(dev)

    def launch():
        _lib.call(name, ptr(gout), ptr(z), ptr(y), ptr(s), ptr(xs), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src,
                  ptr(rev_ptr), ptr(rev_eid), K, 1, m, H, ptr(Q), ptr(P), T, ptr(bufs['Gs']), ptr(bufs['aux']),
                  ptr(bufs['mts']), ptr(bufs['sumG']), ptr(bufs['dz']), ptr(bufs['w']), ptr(bufs['dy_self']), ptr(bufs['dy']),
                  ptr(bufs['dP']), ptr(bufs['dQ']), ptr(ws), wsb, ptr(ticket), st)
    return launch, bufs, ws


new, nb, _ = make('crfconv_meanfield_backward', 'crfconv_meanfield_backward_workspace')
new(); torch.cuda.synchronize()


def finish(bufs):       # dP, dQ for the shapes that leave them to the caller
    if inside:
        return bufs['dP'], bufs['dQ']
    dP = bufs['mts'].view(T * m, H).double().t() @ bufs['Gs'].view(T * m, H).double()
    dQ = z.double().t() @ bufs['sumG'].double()
    return dP.float(), dQ.float()


# float64 reference (torch autograd on the device)
zd, yd, Qd, Pd = (t.double().clone().requires_grad_() for t in (z, y, Q, P))
j = tab.idx32.long()[:, 1:]
d2 = ((yd[:, None, :] - yd[j]) ** 2).sum(-1)
sd = torch.softmax(-d2, dim=1)
x = zd
for t in range(T):
    x = zd @ Qd + (sd[:, :, None] * x[j]).sum(1) @ Pd
x.backward(gout.double())
ref = dict(dz=zd.grad, dy=yd.grad, dQ=Qd.grad, dP=Pd.grad)
for tag, bufs in (('new', nb),):
    dP, dQ = finish(bufs)
    got = dict(dz=bufs['dz'], dy=bufs['dy'], dQ=dQ, dP=dP)
    print(tag + ': ' + '  '.join('%s max|err| %.2e (max|ref| %.2e)' % (k, float((got[k].double() - ref[k]).abs().max()), float(ref[k].abs().max()))
                                 for k in ('dz', 'dy', 'dQ', 'dP')), flush=True)
a0 = [nb[k].clone() for k in ('dz', 'dy', 'dP', 'dQ')]
new(); torch.cuda.synchronize()
print('new bitwise reproducible:', all(torch.equal(a, nb[k]) for a, k in zip(a0, ('dz', 'dy', 'dP', 'dQ'))), flush=True)
print('ticket zero:', int(ops._ticket(dev).abs().sum()) == 0, flush=True)

if not os.environ.get('NOTIME'):
    for rep in range(2):
        for tag, fn in (('new', new),):
            avg, lo = bench._event_time(fn, per=5)
            print('%s backward: avg %.2f us  min %.2f us' % (tag, avg * 1e6, lo * 1e6), flush=True)
