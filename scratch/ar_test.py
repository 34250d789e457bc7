"""Two ranks sharing one GPU over gloo: where does the time go? (diagnostic for the bench's N>1 flow)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from crfconv_amd import distributed as D
rank, world, local = D.init_from_env()
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
flat = torch.randn(820141, device=dev)
a = torch.randn(4096, 4096, device=dev)
def t(fn, n=5):
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(2): dist.all_reduce(flat)
print(rank, 'all_reduce cuda tensor  ms', t(lambda: dist.all_reduce(flat)))
h = flat.cpu()
print(rank, 'all_reduce host tensor  ms', t(lambda: dist.all_reduce(h)))
print(rank, 'matmul 4096 x10 ms', t(lambda: [a @ a for _ in range(10)]))
def both():
    (a @ a); dist.all_reduce(flat); (a @ a)
print(rank, 'matmul+ar+matmul ms', t(both))
dist.barrier(); dist.destroy_process_group()
