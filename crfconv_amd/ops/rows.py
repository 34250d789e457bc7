"""Residual join, LeakyReLU, row gather, neighbour max-pool."""
import ctypes

import torch

from .. import _lib
from ..graph import NeighborTable, ptr, require_gpu, stream_ptr
from ._base import _f32c, _pad_channels

# ------------------------------------------------------------------------------ residual join
class _AddLRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, slope):
        require_gpu(a, b)
        a, b = _f32c(a), _f32c(b)
        out = torch.empty_like(a)
        _lib.call('crfconv_add_lrelu', ptr(a), ptr(b), a.numel(), float(slope), ptr(out), stream_ptr())
        ctx.save_for_backward(out)
        ctx.slope = float(slope)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = _f32c(g)
        gin = torch.empty_like(out)
        _lib.call('crfconv_add_lrelu_backward', ptr(g), ptr(out), out.numel(), ctx.slope, ptr(gin), stream_ptr())
        return gin, gin, None


def add_lrelu(a, b, slope=0.01):
    """leaky_relu(a + b, slope) in one pass (ResNet residual join) on float32 CUDA tensors of equal shape (any element count: the
    flat vectors are padded to the kernel's 16-byte granularity when needed)."""
    require_gpu(a, b)
    if a.shape != b.shape or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise _lib.CrfConvError('add_lrelu: two float32 tensors of one shape (got %s %s, %s %s)' % (tuple(a.shape), a.dtype, tuple(b.shape), b.dtype))
    if a.numel() == 0:
        return a.clone()
    if a.numel() % 4:
        n, shape = a.numel(), a.shape
        pad = 4 - n % 4
        fa, fb = torch.nn.functional.pad(a.reshape(-1), (0, pad)), torch.nn.functional.pad(b.reshape(-1), (0, pad))
        return _AddLRelu.apply(fa, fb, slope)[:n].reshape(shape)
    return _AddLRelu.apply(a, b, slope)


class _LRelu(torch.autograd.Function):
    """leaky_relu(x, slope) alone (a sparse-network Sequential whose Linear has no BatchNorm): the join kernel with a zero addend
    would read a second array; this is the backward kernel's mask applied forward (out = x * lrelu'(x))."""

    @staticmethod
    def forward(ctx, x, slope):
        require_gpu(x)
        x = _f32c(x)
        n = x.numel()
        out = torch.empty_like(x)
        if n % 4 or n == 0:
            raise _lib.CrfConvError('leaky_relu: element count %d must be a positive multiple of 4' % n)
        _lib.call('crfconv_add_lrelu_backward', ptr(x), ptr(x), n, float(slope), ptr(out), stream_ptr())
        ctx.save_for_backward(out)
        ctx.slope = float(slope)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = _f32c(g)
        gin = torch.empty_like(out)
        _lib.call('crfconv_add_lrelu_backward', ptr(g), ptr(out), out.numel(), ctx.slope, ptr(gin), stream_ptr())
        return gin, None


def leaky_relu(x, slope=0.01):
    """F.leaky_relu(x, slope) on a float32 CUDA tensor (element count a multiple of 4) as one library launch."""
    return _LRelu.apply(x, slope)


# ------------------------------------------------------------------------------ gather / max-pool
class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table):
        require_gpu(x)
        x = _f32c(x)
        C = x.shape[1]
        out = torch.empty((table.m_tgt, C), dtype=torch.float32, device=x.device)
        _lib.call('crfconv_gather_rows', ptr(x), ptr(table.idx32), table.m_tgt, C, ptr(out), stream_ptr())
        ctx.table, ctx.m_src = table, x.shape[0]
        return out

    @staticmethod
    def backward(ctx, gout):
        table = ctx.table
        g = _f32c(gout)
        rev_ptr, rev_eid = table.reverse
        dx = torch.empty((ctx.m_src, g.shape[1]), dtype=torch.float32, device=g.device)
        _lib.call('crfconv_gather_rows_backward', ptr(g), ptr(rev_ptr), ptr(rev_eid), ctx.m_src, g.shape[1],
                  ptr(dx), stream_ptr())
        return dx, None


def gather_rows(x, table):
    """out[i] = x[table[i, 0]] (nearest up-sampling); x [m_src, C], C % 4 == 0."""
    if table.K != 1:
        raise ValueError('gather_rows needs a K = 1 table')
    C = x.shape[1]
    Cp = (C + 3) // 4 * 4
    out = _GatherRows.apply(_pad_channels(x, Cp), table)
    return out[:, :C] if Cp != C else out


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table):
        require_gpu(x)
        x = _f32c(x)
        C = x.shape[1]
        out = torch.empty((table.m_tgt, C), dtype=torch.float32, device=x.device)
        arg = torch.empty((table.m_tgt, C), dtype=torch.int32, device=x.device)
        _lib.call('crfconv_neighbor_maxpool_forward', ptr(x), ptr(table.idx32), table.K, table.m_tgt, C, ptr(out),
                  ptr(arg), stream_ptr())
        ctx.table, ctx.m_src = table, x.shape[0]
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, gout):
        (arg,) = ctx.saved_tensors
        table = ctx.table
        g = _f32c(gout)
        rev_ptr, rev_eid = table.reverse
        dx = torch.empty((ctx.m_src, g.shape[1]), dtype=torch.float32, device=g.device)
        _lib.call('crfconv_neighbor_maxpool_backward', ptr(g), ptr(arg), ptr(rev_ptr), ptr(rev_eid), table.K,
                  ctx.m_src, g.shape[1], ptr(dx), stream_ptr())
        return dx, None


def neighbor_maxpool(x, table):
    """out[i, c] = max_k x[table[i, k], c]  (models/point_conv_big.py:74-77)."""
    C = x.shape[1]
    Cp = (C + 3) // 4 * 4
    out = _MaxPool.apply(_pad_channels(x, Cp), table)
    return out[:, :C] if Cp != C else out
