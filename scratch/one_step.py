"""Two eager training steps of the bench configuration (for rocprofv3 --pmc passes: the full bench.py is too long under
counter collection)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, crfconv_amd
from crfconv_amd import models, ops
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
torch.manual_seed(0)
net = models.PointConvBig(6, 13, use_crf=True, steps=3).to(dev).train()
cw = torch.ones(13, device=dev)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for p in net.parameters():
        p.grad = None
    loss = ops.training_loss(net(data), data.y, cw, ignore_index=-1)
    with ops.deferred_weight_grads():
        loss.backward()
torch.cuda.synchronize()
print('done', float(loss), flush=True)
