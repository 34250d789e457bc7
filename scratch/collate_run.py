"""Ten replays of the collate + refresh graph between two marker launches (for rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench, crfconv_amd
from crfconv_amd import models, ops, _lib
from crfconv_amd.ops import ptr, stream_ptr
from crfconv_amd.data import CollateGraph
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
net = models.PointConvBig(6, 13, True, 3).to(dev).train()
loss = ops.training_loss(net(data), data.y, None, ignore_index=-1); loss.backward()
cg = CollateGraph(data, generator=torch.Generator().manual_seed(99))
raw = [bench.synth_cloud(7000 + i, 40960) for i in range(4)]
pos = torch.from_numpy(np.stack([c[0] for c in raw])).to(dev)
x = torch.cat([pos, torch.from_numpy(np.stack([c[1] for c in raw])).to(dev)], -1)
y = torch.from_numpy(np.stack([c[2] for c in raw])).to(dev)
for _ in range(3):
    cg.run(pos, x, y)
torch.cuda.synchronize()
ma, mb, mo = torch.zeros(1, 4, device=dev), torch.zeros(1, 4, device=dev), torch.zeros(1, 8, device=dev)
_lib.call('crfconv_cat2', ptr(ma), ptr(mb), 1, 4, 4, ptr(mo), stream_ptr())       # marker launch (cat2 is not part of the collate)
for _ in range(10):
    cg.graph.replay()
_lib.call('crfconv_cat2', ptr(ma), ptr(mb), 1, 4, 4, ptr(mo), stream_ptr())
torch.cuda.synchronize()
print('done')
