import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench, crfconv_amd
from crfconv_amd import models, ops
from crfconv_amd.data import multiscale_compute
dev = torch.device('cuda', 0)
B, N = 2, 4096
mode = os.environ['MODE']
gen = torch.Generator().manual_seed(1)
data, _ = bench.make_batch(0, B, N, dev, gen, 'morton')
net = models.PointConvBig(6, 13, True, 3).to(dev).train()
ops.training_loss(net(data), data.y, None, ignore_index=-1).backward()
raw = [bench.synth_cloud(70 + i, N) for i in range(B)]
pos = torch.from_numpy(np.stack([c[0] for c in raw])).to(dev)
x = torch.cat([pos, torch.from_numpy(np.stack([c[1] for c in raw])).to(dev)], -1)
y = torch.from_numpy(np.stack([c[2] for c in raw])).to(dev)
sizes = [l.pos.shape[1] for l in data.multiscale]
choices = [torch.randperm(n)[: n // r].sort().values.to(dev) for n, r in zip(sizes, (4, 4, 4, 4, 2))]
fresh = multiscale_compute(pos, x=x, y=y, choices=choices, sort='morton')
def work():
    if mode == 'collate_morton': multiscale_compute(pos, x=x, y=y, choices=choices, sort='morton')
    elif mode == 'collate_nosort': multiscale_compute(pos, x=x, y=y, choices=choices, sort='none')
    elif mode == 'knn_only':
        from crfconv_amd.utils import nearest_neighbors
        nearest_neighbors.knn_batch_device(pos, pos, 16)
    elif mode == 'morton_only':
        from crfconv_amd.data import morton_order
        morton_order(pos)
    elif mode == 'load_only': data.load_(fresh)
    elif mode == 'copies_only':
        for a, b in zip(data.multiscale, fresh.multiscale):
            a.pos.copy_(b.pos); a.neighbor_idx.copy_(b.neighbor_idx)
    elif mode == 'refresh_only':
        from crfconv_amd.graph import table_of
        t = table_of(data.multiscale[0].neighbor_idx, sizes[0]); t.refresh_(fresh.multiscale[0].neighbor_idx)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): work()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): work()
scr = torch.zeros(1000, device=dev)
for i in range(4):
    scr.add_(1.0)                      # an eager launch between replays
    g.replay(); torch.cuda.synchronize()
print(mode, 'OK (4 replays)', flush=True)
