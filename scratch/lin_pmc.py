"""One per-point Linear shape, 10 launches (for rocprofv3 --pmc passes): python3 scratch/lin_pmc.py M Ci Co"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crfconv_amd import ops
M, Ci, Co = (int(v) for v in sys.argv[1:4])
dev = torch.device('cuda', 0)
x = torch.randn(M, Ci, device=dev); W = torch.randn(Co, Ci, device=dev)
for _ in range(10):
    ops._mfma_matmul(x, W, None, False, False)
torch.cuda.synchronize()
