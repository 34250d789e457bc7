"""Level-0 (argv[2]: another level) mean-field forward + backward, N calls in each forward form (argv[1], default 5) (for the rocprofv3 --pmc passes of scratch/pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from crfconv_amd import ops
from crfconv_amd.graph import table_of
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # decoder level: 0 (the roofline's subject), 1, 2, 3
npc = 40960 >> (2 * lvl)
tab = table_of(data.multiscale[lvl].neighbor_idx, npc); tab.reverse
g = torch.Generator().manual_seed(1)
m, H, T = 4 * npc, 8 << lvl, 3
z = torch.randn(m, H, generator=g).to(dev).requires_grad_()
y = (0.5 * torch.randn(m, H, generator=g)).to(dev).requires_grad_()
c = (torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)).to(dev).requires_grad_()
gout = torch.randn(m, H, generator=g).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for mode in ('auto', 'off'):          # auto: the forward as ONE launch with block-resident rows (the table is local); off: one launch per step
    ops.state.mf_block = mode
    for it in range(n):
        for t in (z, y, c): t.grad = None
        ops.crf_meanfield(z, y, c, tab, T).backward(gout)
torch.cuda.synchronize()
print('done')
