"""N eager training steps of the bench configuration (argv[1]) for rocprofv3 --pmc passes over the WHOLE step: the same calls as
bench.py's part_a + part_b (zero grads, forward, weighted CE, backward into the flat bucket, pack, SGD), not captured."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, crfconv_amd
from crfconv_amd import models, ops, distributed as D
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(1234)
data, _ = bench.make_batch(0, 4, 40960, dev, gen, 'morton')
torch.manual_seed(0)
net = models.PointConvBig(6, 13, use_crf=True, steps=3).to(dev).train()
bucket = D.FlatGradAllReduce(net)
opt = crfconv_amd.optim.FlatSGD(bucket, lr=1e-2, momentum=0.95, weight_decay=1e-4)
cw = torch.ones(13, device=dev)
unit = torch.ones((), device=dev)
for it in range(int(sys.argv[1])):
    opt.zero_grad()
    loss = ops.training_loss(net(data), data.y, cw, ignore_index=-1)
    with ops.deferred_weight_grads(sink=bucket.view_of):
        loss.backward(unit)
    bucket.pack()
    opt.step()
torch.cuda.synchronize()
print('done', float(loss), flush=True)
