"""Batch-sharded data parallelism: one process per GPU, full model replica, local BatchNorm
statistics, ONE flat fp32 gradient all-reduce per step over RCCL (torch.distributed backend 'nccl'
is RCCL on ROCm) -- or gloo on CPU in tests.  The reference has no distributed code at all
(trainval.py:24 pins one device); this is the north_star's multi-GPU requirement."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, use_gpu=True):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract). Returns (rank, world, local_rank).
    use_gpu=False: a CPU-only group (gloo) that never touches the HIP runtime -- the many-rank rehearsal of bench.py."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    backend = backend or os.environ.get('CRFCONV_DIST_BACKEND')       # e.g. gloo: several ranks sharing one GPU in tests
    if not use_gpu:
        backend = 'gloo'
    elif backend == 'gloo' and torch.cuda.is_available():
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


def _copy_batched(dsts, srcs):
    """dsts[i].copy_(srcs[i]) for contiguous CUDA tensors of equal dtype: one library launch (crfconv_copy_jobs, graph.hip)."""
    import ctypes
    from . import _lib
    from .graph import stream_ptr
    jobs = (_lib.CopyJob * len(dsts))(*[_lib.CopyJob(s_.data_ptr(), d.data_ptr(), d.numel() * d.element_size())
                                        for d, s_ in zip(dsts, srcs)])
    _lib.call('crfconv_copy_jobs', ctypes.cast(jobs, ctypes.c_void_p), len(dsts), stream_ptr())


class FlatGradAllReduce:
    """One contiguous fp32 bucket for the whole model (820 141 floats = 3.3 MB for PointConvBig(6, 13)):
    a single all-reduce per step.  ``zero()`` sets the gradients to None, so backward installs fresh tensors
    instead of launching one accumulate kernel per parameter (426 launches a step otherwise); ``pack()`` --
    called by ``allreduce_mean()`` at EVERY world size -- copies them into the bucket with one batched foreach
    copy and re-points every ``param.grad`` at its slice of the bucket, which is what ``FlatSGD.step()`` reads.
    The documented step is therefore  zero() -> backward -> allreduce_mean() -> opt.step()  on one GPU too."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        # one slot behind the gradients: the step's failure flag (see publish_guard).  `wire` = what travels in the all-reduce
        self.wire = torch.zeros(n + 1, dtype=torch.float32, device=self.params[0].device)
        self.flat = self.wire[:n]
        self.guard = self.wire[n:]
        self.views = []
        o = 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self._view_of = {id(p): v for p, v in zip(self.params, self.views)}

    def view_of(self, param):
        """The bucket slice of `param` (None for a parameter outside the bucket): pass as
        ``ops.deferred_weight_grads(sink=bucket.view_of)`` and the batched weight-gradient reduction writes straight into
        the bucket -- ``pack()`` then has nothing to copy for those parameters."""
        return self._view_of.get(id(param))

    def zero(self):
        for p in self.params:
            p.grad = None

    def pack(self):
        """param.grad -> bucket (missing gradients count as zero); afterwards every param.grad IS its bucket slice."""
        have = [(v, p.grad) for p, v in zip(self.params, self.views)
                if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        missing = [v for p, v in zip(self.params, self.views) if p.grad is None]
        if have:
            if self.flat.is_cuda and all(g.is_cuda and g.dtype == torch.float32 for _, g in have):
                _copy_batched([v for v, _ in have], [g.contiguous() for _, g in have])
            else:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if missing:
            torch._foreach_zero_(missing)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def publish_guard(self):
        """guard slot <- 1 when a one-launch kernel of this rank's step gave up at its grid barrier (its outputs, hence this
        rank's gradient, are NaN-poisoned), else 0.  Runs in front of the all-reduce, which sums the slot with the gradients: the
        reduced slot is non-zero on EVERY rank when ANY rank failed, and FlatSGD's update kernel skips on it -- all replicas keep
        their parameters and momentum together (the failed rank's NaN is in everybody's bucket by then).  One tiny launch,
        capture-safe; a CPU bucket (gloo tests without a GPU) has no barrier words and publishes 0."""
        if not self.flat.is_cuda:
            self.guard.zero_()
            return
        import ctypes
        from . import _lib, ops
        from .graph import ptr, stream_ptr
        words = ops.fail_word_ptrs(self.flat.device)
        arr = (ctypes.c_void_p * max(len(words), 1))(*words)
        _lib.call('crfconv_sgd_guard_publish', ctypes.cast(arr, ctypes.c_void_p), len(words), ptr(self.guard), stream_ptr())

    def _world(self):
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def allreduce_packed(self):
        """ONE all-reduce (sum) of gradients + guard slot; for a caller that ran pack() and publish_guard() itself (inside its
        captured graph).  Nothing at world size 1."""
        if self._world() > 1:
            dist.all_reduce(self.wire, op=dist.ReduceOp.SUM)

    def allreduce_sum(self):
        """pack() + ONE all-reduce (sum) of the flat bucket.  The mean is taken inside the optimizer's kernel:
        ``FlatSGD(bucket, ..., grad_scale=1 / world_size)`` -- no separate pass over the bucket."""
        self.pack()
        if self._world() > 1:
            self.publish_guard()
            dist.all_reduce(self.wire, op=dist.ReduceOp.SUM)

    def allreduce_mean(self):
        self.pack()
        if self._world() > 1:
            self.publish_guard()
            dist.all_reduce(self.wire, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())        # (the guard slot keeps its sum: any non-zero value means "skip")


def broadcast_parameters(module, src=0):
    """Rank `src`'s parameters and buffers to every rank: ONE broadcast per dtype (values packed into a flat vector and copied
    back), not one per tensor (PointConvBig: ~300)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t.data)
    for dtype, ts in by_dtype.items():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src)
        o = 0
        for t in ts:
            t.copy_(flat[o:o + t.numel()].view_as(t))
            o += t.numel()
