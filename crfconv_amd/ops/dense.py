"""Per-point Linear, BatchNorm step counters, BatchNorm (+ LeakyReLU)."""
import ctypes

import torch

from .. import _lib
from ..graph import NeighborTable, ptr, require_gpu, stream_ptr
from ._base import _f32c, state

# ------------------------------------------------------------------------------ per-point Linear




def _mfma_ok(m, ci, co):
    return m >= state.mfma_min_rows and bool(_lib.load().crfconv_linear_forward_supported(ci, co))


def _gemm(A, B, bias=None, addend=None, nk=False):
    """A [M, K] @ B (+ bias) (+ addend) on the tiled fp32 MFMA kernel of gemm.hip -- the products the row-streaming kernel of
    linear.hip does not take (coarse levels, wide layers; any widths).  nk: B is [N, K] (the F.linear weight), else [K, N].
    Shapes outside the kernel's range (crfconv_gemm_supported: dimensions of 2^24 and more) raise; there is no framework product behind it."""
    M, K = A.shape
    N = B.shape[0] if nk else B.shape[1]
    if M == 0:
        return A.new_empty((0, N))
    if not _lib.load().crfconv_gemm_supported(M, N, K):
        raise _lib.CrfConvError('product %d x %d x %d is outside the tiled kernel\'s range (crfconv_gemm_supported)' % (M, N, K))
    A, B = A.contiguous(), B.contiguous()
    C = torch.empty((M, N), dtype=torch.float32, device=A.device)
    _lib.call('crfconv_gemm', ptr(A), ptr(B), ptr(None if bias is None else bias.contiguous()),
              ptr(None if addend is None else addend.contiguous()), M, N, K, 1 if nk else 0, ptr(C), stream_ptr())
    return C


def _gemm_tn(A, B):
    """A^T B for [m, Ca] / [m, Cb] row operands (a reduction over the long dimension): the MFMA row-reduction kernel of
    linear.hip (crfconv_linear_wgrad), fixed summation order."""
    m, ca = A.shape
    cb = B.shape[1]
    if m == 0:
        return A.new_zeros((ca, cb))
    A, B = A.contiguous(), B.contiguous()
    out = torch.empty((ca, cb), dtype=torch.float32, device=A.device)
    wbytes = _lib.load().crfconv_linear_wgrad_workspace(m, ca, cb)
    wws = torch.empty(wbytes, dtype=torch.uint8, device=A.device)
    _lib.call('crfconv_linear_wgrad', ptr(A), ptr(B), m, ca, cb, ptr(out), None, ptr(wws), wbytes, stream_ptr())
    return out


def _mfma_matmul(x, W, b, transpose_w, want_stats=False):
    """x [m, k] @ (W^T or W) on the fp32 MFMA kernel (linear.hip); optional BatchNorm statistic records."""
    m, ci = x.shape
    co = W.shape[1] if transpose_w else W.shape[0]
    y = torch.empty((m, co), dtype=torch.float32, device=x.device)
    rec = None
    if want_stats:
        nrec = _lib.load().crfconv_linear_forward_stat_records(m)
        rec = torch.empty((nrec, 4, co), dtype=torch.float32, device=x.device)
    _lib.call('crfconv_linear_forward', ptr(x), ptr(W), ptr(b), m, ci, co, 1 if transpose_w else 0, ptr(y), ptr(rec),
              stream_ptr())
    return y, rec


class _Linear(torch.autograd.Function):
    """y = x W^T (+ b) on [m, Ci] rows.  Large-m, <= 128-channel layers run on the MFMA kernels of linear.hip
    (forward with fused BatchNorm statistics, dX, and the dW / db row reduction); small or very wide ones go to
    the vendor GEMM for forward / dX (plain library GEMMs)."""

    @staticmethod
    def forward(ctx, x, W, b, want_stats):
        x = x.contiguous()
        Wc = W.contiguous()
        ctx.save_for_backward(x, Wc)
        # the statistic records are a non-differentiable second output: without this autograd zero-fills a gradient
        # for them on every backward call (33 fill launches per training step of PointConvBig)
        ctx.set_materialize_grads(False)
        ctx.has_bias = b is not None
        ctx.params = (W, b)                      # the parameter objects themselves (deferred weight gradients)
        m, ci = x.shape
        rec = None
        if _mfma_ok(m, ci, Wc.shape[0]):
            y, rec = _mfma_matmul(x, Wc, None if b is None else b.contiguous(), False, want_stats)
        else:
            y = _gemm(x, Wc, b, nk=True)
        if want_stats:
            if rec is None:
                rec = torch.empty(0, device=x.device)
            ctx.mark_non_differentiable(rec)
            return y, rec
        return y

    @staticmethod
    def backward(ctx, g, *_unused):
        if g is None:
            return None, None, None, None
        x, W = ctx.saved_tensors
        g = g.contiguous()
        m, Co = g.shape
        Ci = x.shape[1]
        gx = None
        if ctx.needs_input_grad[0]:
            gx = _mfma_matmul(g, W, None, True)[0] if _mfma_ok(m, Co, Ci) else _gemm(g, W)
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if _defer_ok(ctx.params):
                _defer_weight_grad(g, x, ctx.params, ctx.has_bias)
                return gx, None, None, None
            dW = torch.empty((Co, Ci), dtype=torch.float32, device=g.device)
            db = torch.empty(Co, dtype=torch.float32, device=g.device) if ctx.has_bias else None
            nbytes = _lib.load().crfconv_linear_wgrad_workspace(m, Co, Ci)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
            _lib.call('crfconv_linear_wgrad', ptr(g), ptr(x), m, Co, Ci, ptr(dW), ptr(db), ptr(ws), nbytes,
                      stream_ptr())
        return gx, dW, db, None


def linear(x, W, b=None, want_stats=False):
    """Drop-in for F.linear on [..., Ci] CUDA tensors (CPU tensors are refused: there is no CPU path).  With
    want_stats=True returns (y, records) where `records` feeds bn_act(..., records=records) (empty if unused)."""
    require_gpu(x, W)
    if x.dtype != torch.float32 or W.dtype != torch.float32:
        raise _lib.CrfConvError('linear: float32 only (got %s x %s): the path computes in the reference\'s arithmetic' % (x.dtype, W.dtype))
    shape = x.shape
    out = _Linear.apply(x.reshape(-1, shape[-1]), W, b, want_stats)
    if want_stats:
        y, rec = out
        return y.reshape(shape[:-1] + (W.shape[0],)), (rec if rec.numel() else None)
    return out.reshape(shape[:-1] + (W.shape[0],))


# ------------------------------------------------------------------------------ BatchNorm step counters
_COUNTERS_ADVANCED = False


def tick(bn):
    """num_batches_tracked += 1 of one BatchNorm (torch.nn.BatchNorm1d.forward does this in training)."""
    if not _COUNTERS_ADVANCED and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1


class advance_counters:
    """``with advance_counters(model):`` around a training forward: the num_batches_tracked buffers of every
    BatchNorm in `model` become views of one int64 vector that advances with a single launch (71 one-element
    launches a step for PointConvBig otherwise).  state_dict keys, shapes and values are unchanged."""

    def __init__(self, module):
        self.module = module

    def __enter__(self):
        global _COUNTERS_ADVANCED
        mod = self.module
        cache = mod.__dict__.get('_bn_counter_cache')
        if cache is None:
            bns = [m for m in mod.modules()
                   if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None]
            cache = mod.__dict__['_bn_counter_cache'] = [bns, None]
        bns, flat = cache
        if bns:
            last = bns[-1].num_batches_tracked
            if flat is None or last.device != flat.device or last.data_ptr() != flat[-1].data_ptr():
                flat = cache[1] = torch.stack([b.num_batches_tracked.reshape(()) for b in bns])
                for i, b in enumerate(bns):
                    b._buffers['num_batches_tracked'] = flat[i]
            if flat.is_cuda:
                _lib.call('crfconv_add_i64', ptr(flat), flat.numel(), 1, stream_ptr())
            else:
                flat += 1
        self.prev = _COUNTERS_ADVANCED
        _COUNTERS_ADVANCED = True
        return self

    def __exit__(self, *exc):
        global _COUNTERS_ADVANCED
        _COUNTERS_ADVANCED = self.prev
        return False


# ------------------------------------------------------------------------------ BatchNorm (+ LeakyReLU)
class _BNAct(torch.autograd.Function):
    """y = lrelu(BatchNorm(x), slope) over rows [m, C]: one stats pass + one fused apply pass forward, one
    reduction + one fused pass backward (csrc/bn.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, momentum, eps, use_batch, slope, records):
        m, C = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        coef = torch.empty(4 * C, dtype=torch.float32, device=x.device)
        g, b = _f32c(gamma), _f32c(beta)
        if use_batch and records is not None:
            # statistics came out of the Linear kernel's epilogue: no pass over x for them
            _lib.call('crfconv_bn_apply_from_records', ptr(records), records.shape[0], ptr(x), m, C, ptr(g), ptr(b), ptr(run_mean),
                      ptr(run_var), float(momentum), float(eps), None, float(slope), ptr(coef), ptr(y), stream_ptr())
        else:
            nbytes = _lib.load().crfconv_bn_workspace(m, C)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            _lib.call('crfconv_bn_forward', ptr(x), m, C, ptr(g), ptr(b), ptr(run_mean), ptr(run_var), float(momentum),
                      float(eps), 1 if use_batch else 0, float(slope), ptr(coef), ptr(y), ptr(ws), nbytes, stream_ptr())
        ctx.save_for_backward(x, coef)
        ctx.use_batch, ctx.slope = use_batch, slope
        return y

    @staticmethod
    def backward(ctx, gy):
        x, coef = ctx.saved_tensors
        m, C = x.shape
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        nbytes = _lib.load().crfconv_bn_workspace(m, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        _lib.call('crfconv_bn_backward', ptr(gy), ptr(x), ptr(coef), m, C, 1 if ctx.use_batch else 0, float(ctx.slope),
                  ptr(gx), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, stream_ptr())
        return gx, dgamma, dbeta, None, None, None, None, None, None, None


def bn_act(x, bn, training, slope=1.0, records=None):
    """FastBatchNorm1d semantics (statistics over every leading dim of x [..., C]) fused with LeakyReLU(slope);
    `bn` is the torch.nn.BatchNorm1d holding the affine parameters and running statistics.  A channel count that is not a
    multiple of 4 (the kernels' 16-byte granularity; e.g. a 13-class layer) is zero-padded: the pad channels normalise to
    beta = 0 and are cut off again, the running statistics of the real channels are updated in place."""
    shape = x.shape
    C = shape[-1]
    use_batch = training or bn.running_mean is None
    if training:
        tick(bn)
    mom = 0.1 if bn.momentum is None else bn.momentum
    keep_stats = training or not use_batch
    rm, rv = (bn.running_mean, bn.running_var) if keep_stats else (None, None)
    if C % 4:
        Cp = (C + 3) // 4 * 4
        pad = lambda t, v: None if t is None else torch.nn.functional.pad(t, (0, Cp - C), value=v)
        rmp, rvp = pad(rm, 0.0), pad(rv, 1.0)
        y = _BNAct.apply(pad(x.reshape(-1, C), 0.0), pad(bn.weight, 1.0), pad(bn.bias, 0.0), rmp, rvp, mom, bn.eps, use_batch, slope, None)
        if training and rm is not None:
            with torch.no_grad():
                rm.copy_(rmp[:C])
                rv.copy_(rvp[:C])
        return y[:, :C].reshape(shape)
    y = _BNAct.apply(x.reshape(-1, C), bn.weight, bn.bias, rm, rv, mom, bn.eps, use_batch, slope, records)
    return y.reshape(shape)


# names of the sibling modules, imported LAST: every use is inside a function body, so import cycles between the families are harmless
from .defer import _defer_ok, _defer_weight_grad  # noqa: E402
