import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np, torch
import _seeded as S
from crfconv_amd import models, ops
from crfconv_amd.models.continuous_crf_conv_big import ContinuousGaussianCRFConv as CRF
from crfconv_amd.graph import table_of
from crfconv_amd.utils import nearest_neighbors
dev = torch.device('cuda', 0)
for (N, Nc, U, P_, O, T) in ((2048, 512, 64, 32, 32, 3), (32, 16, 512, 256, 256, 3), (128, 32, 256, 128, 128, 3), (512, 128, 128, 64, 64, 1)):
    g = torch.Generator().manual_seed(N)
    B = 2
    pos = torch.rand(B, N, 3, generator=g).to(dev)
    nbr = nearest_neighbors.knn_batch_device(pos, pos, 16)
    up = torch.randint(0, Nc, (B, N, 1), generator=g).to(dev)
    unary = torch.randn(B, Nc, U, generator=g).to(dev); pair = torch.randn(B, N, P_, generator=g).to(dev)
    layer = CRF(U, P_, O, steps=T)
    layer.load_state_dict(S.fill_state_dict({k: tuple(v.shape) for k, v in layer.state_dict().items()}, 5))
    layer = layer.to(dev).eval()
    with torch.no_grad():
        big = layer(unary, pair, up, nbr)
        parts = [layer(unary[b:b+1].contiguous(), pair[b:b+1].contiguous(), up[b:b+1].contiguous(), nbr[b:b+1].contiguous()) for b in range(B)]
    print('N=%d H=%d T=%d: cloud0 diff %.3e cloud1 diff %.3e (max |out| %.2f)' % (N, O // 4, T, float((big[0] - parts[0][0]).abs().max()), float((big[1] - parts[1][0]).abs().max()), float(big.abs().max())))
