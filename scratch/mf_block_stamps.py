"""Phase stamps (100 MHz s_memrealtime) of the block-resident mean-field forward per workgroup: where the time of the one-launch form goes.
usage: python3 scratch/mf_block_stamps.py [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from crfconv_amd import _lib
from crfconv_amd.graph import ptr, stream_ptr
from crfconv_amd.ops._base import gridsync_ws
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda', 0)
data, _ = bench.make_batch(0, 4, 40960, dev, torch.Generator().manual_seed(1234), 'morton')
tab, m, K, z, y, Q, P, _ = bench._meanfield_problem(data, dev, 8)
H = 8
ws = gridsync_ws(dev)
s_ = torch.empty(m, K, device=dev); xs = torch.empty(T, m, H, device=dev)
for shape in (0, 1):
    rows = 640
    nblk = -(-m // rows)
    dbg = torch.zeros(nblk, 64, dtype=torch.int64, device=dev)
    for it in range(6):
        dbg.zero_()
        _lib.call('crfconv_meanfield_forward_block_stamps', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src, m, ptr(Q), ptr(P), T,
                  ptr(s_), ptr(xs), ptr(ws), shape, ptr(dbg), stream_ptr())
        torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)
    t0 = d[:, 0].min()
    def line(name, col):
        v = (d[:, col] - t0) / 100.0
        print('  %-44s min %6.2f  median %6.2f  max %6.2f us' % (name, v.min(), np.median(v), v.max()))
    print('shape %d (%s): %d rows per workgroup, %d workgroups, T = %d, fail word %d (thread 0 of every workgroup, 100 MHz clock, relative to the first start)'
          % (shape, ('8 wavefronts x 5 half passes', '10 wavefronts x 2 passes')[shape], rows, nblk, T, int(ws[_lib.load().crfconv_gridsync_fail_word()])))
    line('start', 0); line('own rows staged (sync)', 1); line('similarity + step 1 computed', 2); line('x_1 in LDS (two syncs later)', 3)
    for t in range(1, T):
        line('step %d: in-block sums of pass 0 done' % (t + 1), 8 * t); line('step %d: x_%d drained, arrive' % (t + 1, t), 8 * t + 4)
        line('step %d: in-block sums of pass 1 done' % (t + 1), 8 * t + 5); line('step %d: barrier left' % (t + 1), 8 * t + 1)
        line('step %d: out-of-block part + stores issued' % (t + 1), 8 * t + 2)
    line('end (stores drained)', 7)
