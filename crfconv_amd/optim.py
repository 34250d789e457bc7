"""Flat-buffer SGD for the training step of trainval.py:69-73,105 (torch.optim.SGD with momentum and weight decay,
wrapped by ExponentialLR).

All parameters of the model become views of ONE contiguous float32 vector, their gradients views of the
``FlatGradAllReduce`` bucket, and the update is a single kernel launch (csrc/loss.hip: sgd_hyper_kernel) instead of the
~15 multi-tensor launches of the framework optimizer.  Same arithmetic, same state (one momentum buffer).

It IS a ``torch.optim.Optimizer``: ``param_groups[0]['lr']`` is the learning rate, so
``torch.optim.lr_scheduler.ExponentialLR(opt, gamma)`` (trainval.py:73) wraps it unchanged.  The kernel reads
{lr, momentum, dampening, weight_decay} from a 4-float DEVICE tensor that ``step()`` refreshes whenever the group's
values changed; when ``step()`` has been captured into a hipGraph the host code no longer runs, so call
``push_hyper()`` after ``scheduler.step()`` (outside the graph) and the next replay uses the new rate.

Difference from torch.optim.SGD, by construction of the flat vector: weight decay and momentum are applied to every
element each step, also to a parameter whose ``.grad`` was None that step (torch skips such a parameter).  Every
parameter of the networks in crfconv_amd.models receives a gradient each step, so the two agree there
(tests/test_gpu_model.py::test_flat_sgd_matches_torch_sgd)."""
import torch

from . import _lib
from .graph import ptr, require_gpu, stream_ptr


class FlatSGD(torch.optim.Optimizer):
    def __init__(self, bucket, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, check_every=64, grad_scale=1.0):
        """`bucket`: distributed.FlatGradAllReduce of the model (owns the flat gradient vector).  check_every: eager steps
        between two host reads of the grid-barrier failure flag (ops.check_gridsync; 0 = never -- the update kernel itself
        reads the flag every step and skips the update while it is set, so nothing is lost in between).  grad_scale: factor on the
        gradient inside the update -- 1 / world size after ``bucket.allreduce_sum()`` (no separate averaging pass)."""
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        self.bucket = bucket
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
        self.buf = torch.zeros_like(self.flat)
        self.steps = 0
        self.check_every = int(check_every)
        self.grad_scale = float(grad_scale)
        super().__init__(params, dict(lr=float(lr), momentum=float(momentum), dampening=float(dampening),
                                      weight_decay=float(weight_decay), nesterov=bool(nesterov)))
        self._hyper = torch.zeros(5, dtype=torch.float32, device=self.flat.device)
        self._hyper_host = None
        self.push_hyper()

    # convenience mirrors of the single parameter group
    @property
    def lr(self):
        return self.param_groups[0]['lr']

    @lr.setter
    def lr(self, value):
        self.param_groups[0]['lr'] = float(value)

    def add_param_group(self, group):
        if getattr(self, 'param_groups', None):
            raise _lib.CrfConvError('FlatSGD keeps ONE parameter group (one flat vector, one launch)')
        super().add_param_group(group)

    def push_hyper(self):
        """Copies the group's {lr, momentum, dampening, weight_decay} (and grad_scale) to the device block the kernel reads, if they
        changed.  step() does this itself when run eagerly; call it explicitly between replays of a captured step."""
        g = self.param_groups[0]
        cur = (float(g['lr']), float(g['momentum']), float(g['dampening']), float(g['weight_decay']), float(self.grad_scale))
        if cur != self._hyper_host:
            self._hyper.copy_(torch.tensor(cur, dtype=torch.float32), non_blocking=False)
            self._hyper_host = cur

    def zero_grad(self, set_to_none=True):
        self.bucket.zero()

    @torch.no_grad()
    def step(self, closure=None):
        """Expects the gradients in bucket.flat: ``bucket.allreduce_mean()`` (any world size) or ``bucket.pack()``
        puts them there after backward."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        capturing = torch.cuda.is_current_stream_capturing()
        first = self.steps == 0 and g['dampening'] != 0
        if capturing and g['dampening'] != 0:
            raise _lib.CrfConvError('FlatSGD: dampening != 0 makes the first step special; capture is not supported')
        if not capturing:
            self.push_hyper()
        from . import ops
        # the update is guarded by EVERY sticky grid-barrier failure word of this device (the stream this step runs on, other
        # eager streams, the buffer captured graphs use -- an eager step() behind a captured forward / backward sees that one too)
        # and by the bucket's guard slot, which under data parallelism holds the ranks' REDUCED flag (FlatGradAllReduce.
        # publish_guard): a step in which a one-launch kernel timed out anywhere (NaN-poisoned outputs -> NaN gradient, summed into
        # every rank's bucket) leaves parameters and momentum untouched on ALL ranks, in eager steps and captured replays alike,
        # until check_gridsync reports it
        import ctypes
        ops.gridsync_ws(self.flat.device)                    # (creates this stream's words and the capture buffer on first use)
        words = ops.fail_word_ptrs(self.flat.device)
        arr = (ctypes.c_void_p * max(len(words), 1))(*words)
        guard = getattr(self.bucket, 'guard', None)
        _lib.call('crfconv_sgd_step_guarded_all', ptr(self.flat), ptr(self.bucket.flat), ptr(self.buf), self.flat.numel(),
                  ptr(self._hyper), 1 if g['nesterov'] else 0, 1 if first else 0, ctypes.cast(arr, ctypes.c_void_p), len(words),
                  ptr(guard), stream_ptr())   # zero buffer: mu * 0 + g = g
        self.steps += 1
        if not capturing and self.check_every > 0 and self.steps % self.check_every == 0:
            # raises on EVERY rank when any rank failed (the reduced slot); the guarded updates since the failure changed nothing
            ops.check_gridsync(self.flat.device, reduced_flag=getattr(self.bucket, 'guard', None))
        return loss

    def state_dict(self):
        sd = super().state_dict()
        sd['flat_momentum'] = self.buf.clone()
        sd['flat_steps'] = self.steps
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        buf = sd.pop('flat_momentum', None)
        self.steps = int(sd.pop('flat_steps', self.steps))
        super().load_state_dict(sd)
        if buf is not None:
            self.buf.copy_(buf)
        self._hyper_host = None
        self.push_hyper()
