"""kNN call time (graph replays) on the collate's level-0 / level-1 / up-sampling shapes, with the sha1 of the tables.  (Written
for the shared-tile kernel experiment of profiles/r3g_knn_tile_ab.md; the CRFCONV_KNN_TILE_* switches of scratch/run_knn_tile.sh
belonged to that build and are ignored by the shipped library.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, hashlib
import bench
from crfconv_amd.utils import nearest_neighbors as nn_
from crfconv_amd.data import morton_order, pick_rows
dev = torch.device('cuda', 0)
clouds = [bench.synth_cloud(i, 40960) for i in range(4)]
pos = torch.from_numpy(np.stack([c[0] for c in clouds])).to(dev)
pos = pick_rows([pos], morton_order(pos), True)[0]
g = torch.Generator().manual_seed(3)
ch = torch.randperm(40960, generator=g)[:10240].sort().values.to(dev)
sub = pos[:, ch].contiguous()
ch2 = torch.randperm(10240, generator=g)[:2560].sort().values.to(dev)
sub2 = sub[:, ch2].contiguous()
def timeit(fn, n=10):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): out = fn()
    gr.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): gr.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3, out
for name, p, q, K in (('level0 self K16', pos, pos, 16), ('level1 self K16', sub, sub, 16), ('level2 self K16', sub2, sub2, 16),
                      ('up0 (10240 pts, 40960 q) K1', sub, pos, 1), ('up1 (2560 pts, 10240 q) K1', sub2, sub, 1)):
    t, out = timeit(lambda: nn_.knn_batch_device(p, q, K))
    miss = float((out[..., 0] < 0).float().mean())
    print('%-34s %8.1f us per call   sha %s   unanswered %.4f' % (name, t, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12], miss), flush=True)
