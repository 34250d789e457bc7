"""How many of the block-resident forward's 15 out-of-block gather instructions per wavefront-pass carry a request (level-0 table of the
bench batch, blocks of 640 consecutive rows, wavefront-passes of 32 consecutive rows): as the columns stand, and if every lane's share of 8
columns had its out-of-block neighbours moved to one end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
idx = data.multiscale[0].neighbor_idx.reshape(-1, 16).long()            # [m, 16] per-cloud ids
m = idx.shape[0]
npc = 40960
rows = torch.arange(m, device=dev)
glob = idx + (rows // npc * npc)[:, None]
PB = 640
oob = (glob // PB != (rows // PB)[:, None])                              # [m, 16]
oob[:, 0] = False
print('out-of-block fraction of columns 1..15: %.3f' % float(oob[:, 1:].float().mean()))
w = oob[: m // 32 * 32].reshape(-1, 32, 16)                              # wavefront-passes of 32 consecutive rows
now = w.any(1)[:, 1:].sum(1).float()                                     # columns with at least one out-of-block lane
s0 = w[:, :, :8].sum(2).max(1).values.float()                            # after a per-share partition: the longest list per share
s1 = w[:, :, 8:].sum(2).max(1).values.float()
allp = w.sum(2).max(1).values.float()                                    # after a partition over all 15 columns
print('gather instructions with a request per wavefront-pass: now %.2f of 15; per-share partition %.2f; whole-row partition %.2f' % (
    float(now.mean()), float((s0 + s1).mean()), float(allp.mean())))
print('wavefront-passes with no out-of-block neighbour at all: %.3f' % float((now == 0).float().mean()))
for q in (0.25, 0.5, 0.75, 0.9):
    print('  quantile %.2f: now %d, per-share %d, whole-row %d' % (q, int(now.quantile(q)), int((s0 + s1).quantile(q)), int(allp.quantile(q))))
