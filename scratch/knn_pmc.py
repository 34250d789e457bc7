"""Level-0 self-kNN of the bench clouds, 5 calls (for rocprofv3 passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from crfconv_amd.utils import nearest_neighbors as nn_
from crfconv_amd.data import morton_order
dev = torch.device('cuda', 0)
clouds = [bench.synth_cloud(i, 40960) for i in range(4)]
pos = torch.from_numpy(np.stack([c[0] for c in clouds])).to(dev)
order = morton_order(pos)
pos = torch.gather(pos, 1, order.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
for _ in range(5):
    out = nn_.knn_batch_device(pos, pos, 16)
torch.cuda.synchronize()
print('done')
