"""kNN query time vs grid resolution (CRFCONV_KNN_PPC = target points per cell) on the bench's clouds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from crfconv_amd.utils import nearest_neighbors as nn_
dev = torch.device('cuda', 0)
clouds = [bench.synth_cloud(i, 40960) for i in range(4)]
pos = torch.from_numpy(np.stack([c[0] for c in clouds])).to(dev)
if os.environ.get('KNN_SORTED', '1') == '1':      # Morton order, as the device collate hands them to the kNN
    from crfconv_amd.data import morton_order
    order = morton_order(pos)
    pos = torch.gather(pos, 1, order.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
for N in (40960, 10240, 2560):
    p = pos[:, ::40960 // N].contiguous()          # every (40960/N)-th point: still spatially sorted
    for K in (16, 1):
        for _ in range(3): out = nn_.knn_batch_device(p, p, K)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): out = nn_.knn_batch_device(p, p, K)
        b.record(); torch.cuda.synchronize()
        print('PPC %s  N %6d K %2d: %8.1f us per call  checksum %d' % (os.environ.get('CRFCONV_KNN_PPC', '1'), N, K, a.elapsed_time(b) / 20 * 1e3, int(out.sum())), flush=True)
gen = torch.Generator().manual_seed(1)
bench.make_batch(0, 4, 40960, dev, gen, 'morton')
t = [bench.make_batch(0, 4, 40960, dev, gen, 'morton')[1] for _ in range(5)]
print('PPC %s  multiscale_compute %.2f ms' % (os.environ.get('CRFCONV_KNN_PPC', '1'), min(t) * 1e3))
