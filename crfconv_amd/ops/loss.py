"""Weighted soft-max cross-entropy (trainval.py:101-104)."""
import ctypes

import torch

from .. import _lib
from ..graph import NeighborTable, ptr, require_gpu, stream_ptr
from ._base import _f32c, _ticket

# ------------------------------------------------------------------------------ training loss
class _SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight, ignore_index, label_shift):
        require_gpu(logits, target)
        z = _f32c(logits)
        tgt = target.reshape(-1)
        if tgt.dtype != torch.int64:
            tgt = tgt.long()
        tgt = tgt.contiguous()
        m, C = z.shape
        if tgt.numel() != m:
            raise _lib.CrfConvError('cross_entropy: %d targets for %d rows' % (tgt.numel(), m))
        w = None if weight is None else _f32c(weight)
        if w is not None and w.numel() != C:
            raise _lib.CrfConvError('cross_entropy: %d class weights for %d classes' % (w.numel(), C))
        dev = z.device
        lse = torch.empty(m, dtype=torch.float32, device=dev)
        sums = torch.empty(3, dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        nbytes = _lib.load().crfconv_softmax_ce_workspace(m)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.call('crfconv_softmax_ce_forward', ptr(z), ptr(tgt), ptr(w), m, C, int(ignore_index), int(label_shift),
                  ptr(lse), ptr(sums), ptr(loss), ptr(ws), nbytes, ptr(_ticket(dev)), stream_ptr())
        ctx.save_for_backward(z, tgt, w, lse, sums)
        ctx.args = (int(ignore_index), int(label_shift))
        return loss

    @staticmethod
    def backward(ctx, gloss):
        z, tgt, w, lse, sums = ctx.saved_tensors
        ignore_index, label_shift = ctx.args
        g = _f32c(gloss).reshape(1)
        dz = torch.empty_like(z)
        _lib.call('crfconv_softmax_ce_backward', ptr(z), ptr(tgt), ptr(w), ptr(lse), ptr(sums), ptr(g), z.shape[0],
                  z.shape[1], ignore_index, label_shift, ptr(dz), stream_ptr())
        return dz, None, None, None, None


def cross_entropy(logits, target, weight=None, ignore_index=-100, label_shift=0):
    """F.cross_entropy(logits, target - label_shift, weight=weight, ignore_index=ignore_index) (mean reduction) as
    one fused forward and one backward kernel."""
    return _SoftmaxCE.apply(logits, target, weight, ignore_index, label_shift)


def training_loss(logits, labels, class_weights=None, ignore_index=-1):
    """trainval.py:101-104: labels are 1-based (0 = unlabeled -> class -1 = ignore_index after the shift)."""
    return cross_entropy(logits, labels, class_weights, ignore_index, label_shift=1)
