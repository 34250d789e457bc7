"""Level-0 mean-field forward (m = 4 x 40960, H = 8, K = 16, T = 3 on the bench's Morton-ordered clouds): per-step launches against the
one-launch block-resident form (csrc/crf_block.hip), HIP events as in bench.py; bit-equality checked first.
usage: python3 scratch/mf_block_ab.py [T] [sort]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from crfconv_amd import _lib
from crfconv_amd.graph import ptr, stream_ptr
from crfconv_amd.ops._base import gridsync_ws
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sort = sys.argv[2] if len(sys.argv) > 2 else 'morton'
dev = torch.device('cuda', 0)
data, _ = bench.make_batch(0, 4, 40960, dev, torch.Generator().manual_seed(1234), sort)
tab, m, K, z, y, Q, P, _ = bench._meanfield_problem(data, dev, 8)
H = 8
rows = _lib.load().crfconv_meanfield_forward_block_rows(m, H, K, 1, T)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
_lib.call('crfconv_block_locality', ptr(tab.idx32), m, K, 1, rows, ptr(cnt), stream_ptr())
print('m %d rows/block %d  blocks %d  in-block fraction %.3f' % (m, rows, -(-m // rows), cnt.item() / (m * (K - 1))))
ws = gridsync_ws(dev)
s1, s2 = torch.empty(m, K, device=dev), torch.empty(m, K, device=dev)
x1, x2 = torch.empty(T, m, H, device=dev), torch.empty(T, m, H, device=dev)
st = stream_ptr()
def steps(): _lib.call('crfconv_meanfield_forward_u16', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src, K, 1, m, H, ptr(Q), ptr(P), T, ptr(s1), ptr(x1), st)
def block(): _lib.call('crfconv_meanfield_forward_block', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src, K, 1, m, H, ptr(Q), ptr(P), T, ptr(s2), ptr(x2), ptr(ws), st)
def block_nos(): _lib.call('crfconv_meanfield_forward_block', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src, K, 1, m, H, ptr(Q), ptr(P), 1, None, ptr(x2), ptr(ws), st)
steps(); block(); torch.cuda.synchronize()
err = [float((x1[t] - x2[t]).abs().max() / x1[t].abs().max()) for t in range(T)]
print('s equal %s  x_1 equal %s  max |x_t - per-step| / max |x_t|: %s  fail word %d' % (torch.equal(s1, s2), torch.equal(x1[0], x2[0]), ['%.1e' % e for e in err], int(ws[_lib.load().crfconv_gridsync_fail_word()])))
alg = m * (4 * (K - 1) + 4 * H * (2 * T + 1))
for name, fn in (('per-step', steps), ('block', block), ('per-step', steps), ('block', block)):
    avg, lo = bench._event_time(fn)
    print('%-9s avg %.2f us  min %.2f us  frac %.3f' % (name, avg * 1e6, lo * 1e6, alg / avg / 8e12))
