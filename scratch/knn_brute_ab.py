"""Small-cloud kNN (sixteen lanes per query, no grid) against the grid search on the collate's coarse levels: identical
neighbour tables required, time per call.  The A/B switch is read once per process: run with and without
CRFCONV_KNN_NO_BRUTE=1 and compare the checksums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench, hashlib
from crfconv_amd.utils import nearest_neighbors as nn_
from crfconv_amd.data import morton_order
dev = torch.device('cuda', 0)
clouds = [bench.synth_cloud(i, 40960) for i in range(4)]
pos = torch.from_numpy(np.stack([c[0] for c in clouds])).to(dev)
order = morton_order(pos)
pos = torch.gather(pos, 1, order.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
g = torch.Generator().manual_seed(3)
levels = [pos]
for r in (4, 4, 4, 4):
    n = levels[-1].shape[1]
    ch = torch.randperm(n, generator=g)[: n // r].sort().values.to(dev)
    levels.append(levels[-1][:, ch].contiguous())
for li, p in enumerate(levels):
    for K, q in ((16, p), (1, levels[li - 1] if li > 0 else p)):
        for _ in range(3): out = nn_.knn_batch_device(p, q, K)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): out = nn_.knn_batch_device(p, q, K)
        b.record(); torch.cuda.synchronize()
        print('level %d  points/cloud %6d  queries/cloud %6d  K %2d: %8.1f us per call   sha1 %s' % (
            li, p.shape[1], q.shape[1], K, a.elapsed_time(b) / 20 * 1e3, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]), flush=True)
