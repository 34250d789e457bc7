"""Batch-sharded data parallelism: one process per GPU, full model replica, local BatchNorm
statistics, ONE flat fp32 gradient all-reduce per step over RCCL (torch.distributed backend 'nccl'
is RCCL on ROCm) -- or gloo on CPU in tests.  The reference has no distributed code at all
(trainval.py:24 pins one device); this is the north_star's multi-GPU requirement."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract). Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    backend = backend or os.environ.get('CRFCONV_DIST_BACKEND')       # e.g. gloo: several ranks sharing one GPU in tests
    if backend == 'gloo' and torch.cuda.is_available():
        local %= torch.cuda.device_count()
    # under torchrun (WORLD_SIZE exported) the group is created even for one rank: same code path at every N
    if (world > 1 or 'WORLD_SIZE' in os.environ) and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class FlatGradAllReduce:
    """One contiguous fp32 bucket for the whole model (820 141 floats = 3.3 MB for PointConvBig(6, 13)):
    a single all-reduce per step.  Gradients are left as autograd produces them (``zero()`` sets them to
    None, so backward installs fresh tensors instead of launching one accumulate kernel per parameter --
    426 launches a step otherwise) and are packed / unpacked with batched foreach copies only when there
    is more than one rank."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self.views = []
        o = 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def zero(self):
        for p in self.params:
            p.grad = None

    def allreduce_mean(self):
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(self.views, grads)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())
        for p, v in zip(self.params, self.views):
            p.grad = v


def broadcast_parameters(module, src=0):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)
