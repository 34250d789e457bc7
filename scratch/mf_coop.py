"""EXPERIMENT: the coarse levels' mean-field forward as ONE launch with grid barriers (crfconv_meanfield_forward_coop) against the
T per-step launches: HIP-event time per level and bit-equality of every iterate and of the soft-max weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from crfconv_amd import _lib
from crfconv_amd.graph import ptr, stream_ptr
dev = torch.device('cuda', 0)
data, _ = bench.make_batch(0, 4, 40960, dev, torch.Generator().manual_seed(1234), 'morton')
lib = _lib.load()
T = 3
for level in (1, 2, 3):
    H = 8 << level
    tab, m, K, z, y, Q, P, _ = bench._meanfield_problem(data, dev, H, level=level)
    ok = lib.crfconv_meanfield_coop_supported(m, H, K, 1, T)
    s0, x0 = torch.empty(m, K, device=dev), torch.empty(T, m, H, device=dev)
    s1, x1 = torch.full((m, K), 7.0, device=dev), torch.full((T, m, H), 7.0, device=dev)
    ws = torch.zeros(lib.crfconv_gridsync_workspace(), dtype=torch.uint8, device=dev)
    st = stream_ptr()
    def steps():
        _lib.call('crfconv_meanfield_forward_u16', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src, K, 1, m, H, ptr(Q), ptr(P), T, ptr(s0), ptr(x0), st)
    def coop():
        _lib.call('crfconv_meanfield_forward_coop', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src, K, 1, m, H, ptr(Q), ptr(P), T, ptr(s1), ptr(x1), ptr(ws), ws.numel(), st)
    steps()
    line = 'level %d  H %2d  m %6d  per-step launches %6.2f us' % (level, H, m, bench._event_time(steps)[0] * 1e6)
    if ok == 1:
        coop(); torch.cuda.synchronize()
        eq = bool(torch.equal(s0, s1)) and bool(torch.equal(x0, x1))
        fail = int(ws.view(torch.int32)[lib.crfconv_gridsync_fail_word()].item())
        line += '   one launch + %d grid barriers %6.2f us   bit-identical %s   barrier failure word %d' % (T - 1, bench._event_time(coop)[0] * 1e6, eq, fail)
    else:
        line += '   (one-launch form not co-resident / not supported)'
    print(line, flush=True)
