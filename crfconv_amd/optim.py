"""Flat-buffer SGD for the training step of trainval.py:69-72,105 (torch.optim.SGD with momentum and weight decay).

All parameters of the model become views of ONE contiguous float32 vector, their gradients views of the
``FlatGradAllReduce`` bucket, and the update is a single kernel launch (csrc/loss.hip: sgd_kernel) instead of the
~15 multi-tensor launches of the framework optimizer.  Same arithmetic, same state (one momentum buffer)."""
import torch

from . import _lib
from .graph import ptr, require_gpu, stream_ptr


class FlatSGD:
    def __init__(self, bucket, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        """`bucket`: distributed.FlatGradAllReduce of the model (owns the flat gradient vector)."""
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        self.bucket = bucket
        self.lr, self.momentum, self.dampening = float(lr), float(momentum), float(dampening)
        self.weight_decay, self.nesterov = float(weight_decay), bool(nesterov)
        params = bucket.params
        require_gpu(*params)
        if any(p.dtype != torch.float32 for p in params):
            raise _lib.CrfConvError('FlatSGD: float32 parameters only')
        self.flat = torch.empty_like(bucket.flat)
        o = 0
        for p in params:                      # re-home every parameter inside the flat vector (values preserved)
            view = self.flat[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            o += p.numel()
        self.buf = torch.zeros_like(self.flat) if self.momentum != 0 else None
        self.steps = 0

    def zero_grad(self):
        self.bucket.zero()

    def step(self):
        """Expects the gradients in bucket.flat (bench / trainer copy or all-reduce them there)."""
        _lib.call('crfconv_sgd_step', ptr(self.flat), ptr(self.bucket.flat), ptr(self.buf), self.flat.numel(), self.lr,
                  self.momentum, self.dampening, self.weight_decay, 1 if self.nesterov else 0,
                  1 if (self.steps == 0 and self.dampening != 0) else 0, stream_ptr())   # zero buffer: mu * 0 + g = g
        self.steps += 1
