import torch
a = torch.empty(163840 * 32, device='cuda')
b = torch.empty(163840 * 32, device='cuda')
for _ in range(5):
    a.fill_(1.0)
for _ in range(5):
    b.copy_(a)
torch.cuda.synchronize()
