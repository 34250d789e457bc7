"""Linear -> BatchNorm -> LeakyReLU blocks as single nodes: row-streaming forms, the classifier head, coarse-level forms and groups."""
import ctypes

import torch

from .. import _lib
from ..graph import NeighborTable, ptr, require_gpu, stream_ptr
from ._base import _f32c, _mlp_ticket, _ticket, gridsync_ws, state

# ------------------------------------------------------------------------------ Linear -> BatchNorm -> LeakyReLU as one op


class _MLPBlock(torch.autograd.Function):
    """A = lrelu(BN_train(x W^T), slope) (models/common.py:34-40).  Forward: the MFMA Linear with statistic records in its
    epilogue, coefficients, one fused apply pass.  Backward: crfconv_mlp_backward -- one pass over (gA, y, x) for dgamma,
    dbeta, dW, one pass over (gA, y) for dX; the BatchNorm input gradient never reaches memory.

    fork: the node also returns its input (as an alias) for the input's OTHER consumer -- the shortcut of a ResNet block --
    so that the gradient coming back through the alias reaches this node's backward, which adds it while writing dX
    (crfconv_mlp_backward_add) instead of autograd running an accumulation pass over three [M, Ci] tensors."""

    @staticmethod
    def forward(ctx, x_in, W, gamma, beta, run_mean, run_var, momentum, eps, slope, fork=False):
        x = x_in.contiguous()
        Wc = W.contiguous()
        m, ci = x.shape
        co = Wc.shape[0]
        y, rec = _mfma_matmul(x, Wc, None, False, True)
        coef = torch.empty(4 * co, dtype=torch.float32, device=x.device)
        g, b = _f32c(gamma), _f32c(beta)
        out = torch.empty_like(y)
        # coefficients and apply in one launch (same values)
        _lib.call('crfconv_bn_apply_from_records', ptr(rec), rec.shape[0], ptr(y), m, co, ptr(g), ptr(b), ptr(run_mean), ptr(run_var),
                  float(momentum), float(eps), None, float(slope), ptr(coef), ptr(out), stream_ptr())
        ctx.prm = (W, gamma, beta)
        ctx.save_for_backward(x, Wc, y, coef)
        ctx.slope = float(slope)
        if fork:
            ctx.set_materialize_grads(False)           # an unused alias must not cost a zero fill
            return out, x_in
        return out

    @staticmethod
    def backward(ctx, gA, g_alias=None):
        x, W, y, coef = ctx.saved_tensors
        m, ci = x.shape
        co = W.shape[0]
        dev = x.device
        gA = torch.zeros_like(y) if gA is None else gA.contiguous()
        dX = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        add = _f32c(g_alias) if (g_alias is not None and dX is not None) else None
        outs, dfr = _mlp_param_outs(ctx.prm, W, dev)       # (dW, dgamma, dbeta) targets; dfr: dW finished at the end of the pass
        dW, dgamma, dbeta = (o[0] for o in outs)
        nbytes = _lib.load().crfconv_mlp_backward_workspace(m, ci, co)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.call('crfconv_mlp_backward_add', ptr(gA), ptr(y), ptr(x), ptr(W), ptr(coef), ctx.slope, m, ci, co, ptr(add), ptr(dX),
                  ptr(None if dfr else dW), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, _mlp_ticket(dev), stream_ptr())
        return dX, *_mlp_param_rets(ctx.prm, outs, dfr, ws, m, ci, co, coef), None, None, None, None, None, None




class _MLPBlockJoin(torch.autograd.Function):
    """out = lrelu(BN_train(x W^T) + skip, slope): the tail of a ResNet block (models/point_conv_big.py:84-88: lin_out has no
    activation, then F.leaky_relu(x + shortcut)) as one node -- forward: MFMA Linear with statistic records, coefficients, ONE
    pass for BatchNorm + residual add + LeakyReLU (crfconv_bn_apply_add; the normalised tensor never reaches memory);
    backward: g1 = g lrelu'(out) is both the skip gradient and the gA of crfconv_mlp_backward (BatchNorm without activation)."""

    @staticmethod
    def forward(ctx, x, W, gamma, beta, run_mean, run_var, momentum, eps, skip, slope):
        x, Wc, skip = x.contiguous(), W.contiguous(), skip.contiguous()
        m, ci = x.shape
        co = Wc.shape[0]
        y, rec = _mfma_matmul(x, Wc, None, False, True)
        coef = torch.empty(4 * co, dtype=torch.float32, device=x.device)
        st = stream_ptr()
        out = torch.empty_like(y)
        _lib.call('crfconv_bn_apply_from_records', ptr(rec), rec.shape[0], ptr(y), m, co, ptr(_f32c(gamma)), ptr(_f32c(beta)),
                  ptr(run_mean), ptr(run_var), float(momentum), float(eps), ptr(skip), float(slope), ptr(coef), ptr(out), st)
        ctx.prm = (W, gamma, beta)
        ctx.save_for_backward(x, Wc, y, coef, out)
        ctx.slope = float(slope)
        return out

    @staticmethod
    def backward(ctx, g):
        x, W, y, coef, out = ctx.saved_tensors
        m, ci = x.shape
        co = W.shape[0]
        dev = x.device
        g = g.contiguous()
        g1 = torch.empty_like(g)
        st = stream_ptr()
        _lib.call('crfconv_add_lrelu_backward', ptr(g), ptr(out), out.numel(), ctx.slope, ptr(g1), st)
        dX = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        outs, dfr = _mlp_param_outs(ctx.prm, W, dev)       # (dW, dgamma, dbeta) targets; dfr: dW finished at the end of the pass
        dW, dgamma, dbeta = (o[0] for o in outs)
        nbytes = _lib.load().crfconv_mlp_backward_workspace(m, ci, co)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.call('crfconv_mlp_backward', ptr(g1), ptr(y), ptr(x), ptr(W), ptr(coef), 1.0, m, ci, co, ptr(dX), ptr(None if dfr else dW),
                  ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, _mlp_ticket(dev), st)
        gskip = g1 if ctx.needs_input_grad[8] else None
        return dX, *_mlp_param_rets(ctx.prm, outs, dfr, ws, m, ci, co, coef), None, None, None, None, gskip, None




class _MLPBlockDropout(torch.autograd.Function):
    """dropout(lrelu(BN_train(x W^T), slope), p): the classifier's MLP -> nn.Dropout (models/point_conv_big.py:131-134) as one
    node.  Forward: MFMA Linear with statistic records, coefficients, ONE pass for BatchNorm + LeakyReLU + dropout
    (crfconv_bn_apply_dropout); the mask is a hash of (seed, the BatchNorm's step counter, element index), so nothing is
    stored, forward and backward of a step agree, and a replayed hipGraph draws a new mask every step (the counter is a
    device word that the forward advances).  Backward: the same mask on the incoming gradient, then crfconv_mlp_backward."""

    @staticmethod
    def forward(ctx, x, W, gamma, beta, run_mean, run_var, momentum, eps, slope, p, seed, counter):
        x, Wc = x.contiguous(), W.contiguous()
        m, ci = x.shape
        co = Wc.shape[0]
        y, rec = _mfma_matmul(x, Wc, None, False, True)
        coef = torch.empty(4 * co, dtype=torch.float32, device=x.device)
        st = stream_ptr()
        _lib.call('crfconv_bn_coef_from_records', ptr(rec), m, co, ptr(_f32c(gamma)), ptr(_f32c(beta)), ptr(run_mean),
                  ptr(run_var), float(momentum), float(eps), ptr(coef), st)
        out = torch.empty_like(y)
        used = torch.empty(1, dtype=torch.int64, device=x.device)     # the counter value of THIS call's mask, for its backward
        _lib.call('crfconv_bn_apply_dropout', ptr(y), m, co, ptr(coef), float(slope), float(p), int(seed), ptr(counter), ptr(out),
                  ptr(used), st)
        counter = used
        ctx.prm = (W, gamma, beta)
        ctx.save_for_backward(x, Wc, y, coef, counter)
        ctx.slope, ctx.p, ctx.seed = float(slope), float(p), int(seed)
        return out

    @staticmethod
    def backward(ctx, g):
        x, W, y, coef, counter = ctx.saved_tensors
        m, ci = x.shape
        co = W.shape[0]
        dev = x.device
        g = g.contiguous()
        gA = torch.empty_like(g)
        st = stream_ptr()
        _lib.call('crfconv_dropout_backward', ptr(g), g.numel(), ctx.p, ctx.seed, ptr(counter), ptr(gA), st)
        dX = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        outs, dfr = _mlp_param_outs(ctx.prm, W, dev)       # (dW, dgamma, dbeta) targets; dfr: dW finished at the end of the pass
        dW, dgamma, dbeta = (o[0] for o in outs)
        nbytes = _lib.load().crfconv_mlp_backward_workspace(m, ci, co)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.call('crfconv_mlp_backward', ptr(gA), ptr(y), ptr(x), ptr(W), ptr(coef), ctx.slope, m, ci, co, ptr(dX), ptr(None if dfr else dW),
                  ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, _mlp_ticket(dev), st)
        return dX, *_mlp_param_rets(ctx.prm, outs, dfr, ws, m, ci, co, coef), None, None, None, None, None, None, None, None


class _MLPDropoutLinear(torch.autograd.Function):
    """logits = dropout(lrelu(BN_train(x W1^T), slope), p) W2^T + b2: the whole classifier (models/point_conv_big.py:131-134,
    MLP -> nn.Dropout -> nn.Linear) as one node.  Forward: _MLPBlockDropout's three launches plus the MFMA Linear.  Backward: the
    input gradient of the last Linear is masked WHILE IT IS WRITTEN (crfconv_linear_forward_dropout: same counter-based mask)
    -- the separate dropout-backward pass over [m, 4 C] is gone --, then crfconv_mlp_backward; (dW2, db2) as _Linear."""

    @staticmethod
    def forward(ctx, x, W, gamma, beta, run_mean, run_var, momentum, eps, slope, p, seed, counter, W2, b2):
        x, Wc, W2c = x.contiguous(), W.contiguous(), W2.contiguous()
        m, ci = x.shape
        co = Wc.shape[0]
        y, rec = _mfma_matmul(x, Wc, None, False, True)
        coef = torch.empty(4 * co, dtype=torch.float32, device=x.device)
        st = stream_ptr()
        _lib.call('crfconv_bn_coef_from_records', ptr(rec), m, co, ptr(_f32c(gamma)), ptr(_f32c(beta)), ptr(run_mean),
                  ptr(run_var), float(momentum), float(eps), ptr(coef), st)
        h = torch.empty_like(y)
        used = torch.empty(1, dtype=torch.int64, device=x.device)     # the counter value of THIS call's mask, for its backward
        _lib.call('crfconv_bn_apply_dropout', ptr(y), m, co, ptr(coef), float(slope), float(p), int(seed), ptr(counter), ptr(h),
                  ptr(used), st)
        counter = used
        logits = _mfma_matmul(h, W2c, None if b2 is None else b2.contiguous(), False)[0]
        ctx.prm = (W, gamma, beta)
        ctx.save_for_backward(x, Wc, y, coef, counter, h, W2c)
        ctx.slope, ctx.p, ctx.seed = float(slope), float(p), int(seed)
        ctx.params2 = (W2, b2)
        return logits

    @staticmethod
    def backward(ctx, g):
        x, W, y, coef, counter, h, W2 = ctx.saved_tensors
        m, ci = x.shape
        co, c2 = W.shape[0], W2.shape[0]
        dev = x.device
        g = g.contiguous()
        st = stream_ptr()
        gA = torch.empty((m, co), dtype=torch.float32, device=dev)       # = dropout'(g W2): masked by the kernel that forms it
        _lib.call('crfconv_linear_forward_dropout', ptr(g), ptr(W2), m, c2, co, 1, ctx.p, ctx.seed, ptr(counter), ptr(gA), st)
        W2p, b2p = ctx.params2
        dW2 = db2 = None
        if _defer_ok(ctx.params2):
            _defer_weight_grad(g, h, ctx.params2, b2p is not None)
        else:
            dW2 = torch.empty((c2, co), dtype=torch.float32, device=dev)
            db2 = torch.empty(c2, dtype=torch.float32, device=dev) if b2p is not None else None
            nb = _lib.load().crfconv_linear_wgrad_workspace(m, c2, co)
            wsw = torch.empty(nb, dtype=torch.uint8, device=dev)
            _lib.call('crfconv_linear_wgrad', ptr(g), ptr(h), m, c2, co, ptr(dW2), ptr(db2), ptr(wsw), nb, st)
        dX = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        outs, dfr = _mlp_param_outs(ctx.prm, W, dev)       # (dW, dgamma, dbeta) targets; dfr: dW finished at the end of the pass
        dW, dgamma, dbeta = (o[0] for o in outs)
        nbytes = _lib.load().crfconv_mlp_backward_workspace(m, ci, co)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.call('crfconv_mlp_backward', ptr(gA), ptr(y), ptr(x), ptr(W), ptr(coef), ctx.slope, m, ci, co, ptr(dX), ptr(None if dfr else dW),
                  ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, _mlp_ticket(dev), st)
        return dX, *_mlp_param_rets(ctx.prm, outs, dfr, ws, m, ci, co, coef), None, None, None, None, None, None, None, None, dW2, db2




class _HeadRecompute(torch.autograd.Function):
    """The same classifier -- logits = dropout(lrelu(BN_train(x W1^T), slope), p) W2^T + b2 (models/point_conv_big.py:131-134) --
    without a stored [m, 4 C] tensor (csrc/head.hip): the statistics pass keeps only the BatchNorm records, every later pass
    recomputes x W1^T from the [m, C] input on the matrix pipe, and the dropout mask travels as one bit per element.  Logits are
    bit-identical to _MLPDropoutLinear's (same products in the same order, same mask); the parameter gradients are summed in
    float64 from per-workgroup partials."""

    @staticmethod
    def forward(ctx, x, W, gamma, beta, run_mean, run_var, momentum, eps, slope, p, seed, counter, W2, b2):
        x, Wc, W2c = x.contiguous(), W.contiguous(), W2.contiguous()
        m, ci = x.shape
        co, c2 = Wc.shape[0], W2c.shape[0]
        dev = x.device
        lib = _lib.load()
        st = stream_ptr()
        nrec = lib.crfconv_head_stat_records(m)
        rec = torch.empty((nrec, co, 4), dtype=torch.float32, device=dev)
        _lib.call('crfconv_head_stats', ptr(x), ptr(Wc), m, ci, co, ptr(rec), st)
        coef = torch.empty(4 * co, dtype=torch.float32, device=dev)
        _lib.call('crfconv_bn_coef_from_nrecords', ptr(rec), nrec, m, co, ptr(_f32c(gamma)), ptr(_f32c(beta)), ptr(run_mean),
                  ptr(run_var), float(momentum), float(eps), ptr(coef), st)
        logits = torch.empty((m, c2), dtype=torch.float32, device=dev)
        mask = torch.empty(lib.crfconv_head_mask_words(m), dtype=torch.int32, device=dev)
        b2c = None if b2 is None else b2.contiguous()
        _lib.call('crfconv_head_forward', ptr(x), ptr(Wc), ptr(coef), float(slope), float(p), int(seed), ptr(counter), ptr(W2c),
                  ptr(b2c), m, ci, co, c2, ptr(logits), ptr(mask), None, st)
        ctx.prm = (W, gamma, beta, W2, b2)
        ctx.save_for_backward(x, Wc, coef, mask, W2c)
        ctx.slope, ctx.p = float(slope), float(p)
        return logits

    @staticmethod
    def backward(ctx, g):
        x, W, coef, mask, W2 = ctx.saved_tensors
        m, ci = x.shape
        co, c2 = W.shape[0], W2.shape[0]
        dev = x.device
        g = g.contiguous()
        Wp, gp, bp, W2p, b2p = ctx.prm
        outs = [_param_out(Wp, (co, ci), dev), _param_out(gp, (co,), dev), _param_out(bp, (co,), dev), _param_out(W2p, (c2, co), dev)]
        ob2 = _param_out(b2p, (c2,), dev) if b2p is not None else (None, False)
        dX = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        nbytes = _lib.load().crfconv_head_backward_workspace(m, ci, co, c2)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.call('crfconv_head_backward', ptr(g), ptr(x), ptr(W), ptr(coef), ctx.slope, ctx.p, ptr(W2), ptr(mask), m, ci, co, c2,
                  ptr(dX), ptr(outs[0][0]), ptr(outs[1][0]), ptr(outs[2][0]), ptr(outs[3][0]), ptr(ob2[0]), ptr(ws), nbytes, stream_ptr())
        rets = [_param_ret(prm, o[0], o[1]) for prm, o in zip((Wp, gp, bp, W2p), outs)]
        rb2 = _param_ret(b2p, ob2[0], ob2[1]) if b2p is not None else None
        return dX, rets[0], rets[1], rets[2], None, None, None, None, None, None, None, None, rets[3], rb2


def dropout_seed(ci, co):
    """Seed of the counter-based dropout mask of a fused MLP -> Dropout block with `ci` inputs and `co` outputs (a
    function of torch.initial_seed() and the layer shape)."""
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + co * 7919 + ci) & 0xFFFFFFFFFFFFFFFF


def dropout_keep_mask(seed, counter, n, p):
    """The mask csrc/common.hpp::dropout_keep draws for elements 0 .. n-1 of the call with (seed, counter), evaluated on
    the HOST (numpy bool array, True = kept): two 32-bit keys = the halves of one splitmix64 round of (seed, counter); element e
    is kept iff the keyed lowbias32 hash of e reaches p 2^32.  `counter` = the classifier BatchNorm's num_batches_tracked AFTER
    the forward (the forward advances it before the kernel reads it).  A caller -- the parity tests -- hands the same mask to
    another implementation of the network."""
    import numpy as np
    t = float(p) * 4294967296.0
    thr = np.uint32(0xffffffff if t >= 4294967295.0 else (0 if t <= 0.0 else int(t)))
    mask64 = 0xFFFFFFFFFFFFFFFF
    z = (int(seed) + 0x9E3779B97F4A7C15 * (int(counter) + 1)) & mask64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask64
    z ^= z >> 31
    k0, k1 = np.uint32(z & 0xFFFFFFFF), np.uint32(z >> 32)
    with np.errstate(over='ignore'):
        e = np.arange(int(n), dtype=np.uint64)
        hi = (e >> np.uint64(32)).astype(np.uint32)
        x = e.astype(np.uint32) + k0 + ((hi << np.uint32(13)) | (hi >> np.uint32(19)))
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7feb352d)
        x ^= x >> np.uint32(15)
        x += k1
        x *= np.uint32(0x846ca68b)
        x ^= x >> np.uint32(16)
    return x >= thr


def mlp_dropout_linear(x, W, bn, slope, p, W2, b2, recompute=None):
    """Linear(dropout(lrelu(BatchNorm_train(x W^T), slope), p)) as one node where the fused dropout block and the MFMA Linear
    both apply, else None: _HeadRecompute (no [m, 4 C] tensor is ever stored) for the shapes csrc/head.hip covers, else
    _MLPDropoutLinear.  recompute: None = that choice, False = the stored form, True = insist on the recomputing one."""
    if not (0.0 <= p < 1.0) or bn.num_batches_tracked is None:
        return None
    m = x.numel() // x.shape[-1]
    ci, co, c2 = x.shape[-1], W.shape[0], W2.shape[0]
    if not (mlp_block_ok(x, W, None, bn, True) and not _mlp_small_ok(m, ci, co) and _mfma_ok(m, co, c2) and _mfma_ok(m, c2, co)
            and W2.dtype == torch.float32):
        return None
    require_gpu(x, W, W2)
    tick(bn)
    mom = 0.1 if bn.momentum is None else bn.momentum
    seed = dropout_seed(ci, co)                          # as mlp_block_dropout
    node = _MLPDropoutLinear
    head_ok = bool(_lib.load().crfconv_head_supported(m, ci, co, c2))
    if recompute and not head_ok:
        return None
    if head_ok and (recompute or recompute is None):
        node = _HeadRecompute                            # no [m, 4 C] tensor at all
    out = node.apply(x.reshape(-1, ci), W, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps, slope,
                                  p, seed, bn.num_batches_tracked, W2, b2)
    return out.reshape(x.shape[:-1] + (c2,))


def mlp_block_dropout(x, W, bn, slope, p):
    """dropout(lrelu(BatchNorm_train(x W^T), slope), p) as one node where the big-level fused block applies, else None (the
    caller then runs its own MLP and nn.Dropout).  The mask stream is seeded from torch.initial_seed() and advances with the
    BatchNorm's num_batches_tracked -- reproducible under torch.manual_seed, but NOT the draws nn.Dropout would have made."""
    if not (0.0 <= p < 1.0) or bn.num_batches_tracked is None:
        return None
    m = x.numel() // x.shape[-1]
    ci, co = x.shape[-1], W.shape[0]
    if not mlp_block_ok(x, W, None, bn, True):
        return None
    if _mlp_small_ok(m, ci, co):
        # below the row-streaming forms' switch-over the node still runs on them when they TAKE the shape (they do for every width
        # of the reference networks): the counter-based mask at every size, not the module's nn.Dropout with the framework's draws
        lib = _lib.load()
        if not (co % 4 == 0 and lib.crfconv_linear_forward_supported(ci, co) and lib.crfconv_mlp_backward_supported(m, ci, co) == 1):
            return None
    require_gpu(x, W)
    tick(bn)                                              # advances the counter the mask is keyed on (unless the model already did)
    mom = 0.1 if bn.momentum is None else bn.momentum
    seed = dropout_seed(ci, co)                          # stable per layer shape
    out = _MLPBlockDropout.apply(x.reshape(-1, ci), W, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps, slope,
                                 p, seed, bn.num_batches_tracked)
    return out.reshape(x.shape[:-1] + (co,))


class _MLPBlockPool(torch.autograd.Function):
    """max over the table's neighbours of BN_train(x W^T): the strided shortcut of a ResNet block
    (models/point_conv_big.py:74-83) as one node.  Forward: MFMA Linear with statistic records, coefficients, max-pool that
    applies the BatchNorm affine while gathering (crfconv_neighbor_maxpool_affine_forward) -- the normalised fine-level
    tensor never reaches memory; backward: the pool's scatter gives gA, then crfconv_mlp_backward (BatchNorm, no activation).
    fork: as _MLPBlock -- the node hands its input on as an alias and adds the alias' gradient while writing dX."""

    @staticmethod
    def forward(ctx, x_in, W, gamma, beta, run_mean, run_var, momentum, eps, table, fork=False):
        x, Wc = x_in.contiguous(), W.contiguous()
        m, ci = x.shape
        co = Wc.shape[0]
        dev = x.device
        y, rec = _mfma_matmul(x, Wc, None, False, True)
        coef = torch.empty(4 * co, dtype=torch.float32, device=dev)
        st = stream_ptr()
        _lib.call('crfconv_bn_coef_from_records', ptr(rec), m, co, ptr(_f32c(gamma)), ptr(_f32c(beta)), ptr(run_mean),
                  ptr(run_var), float(momentum), float(eps), ptr(coef), st)
        out = torch.empty((table.m_tgt, co), dtype=torch.float32, device=dev)
        arg = torch.empty((table.m_tgt, co), dtype=torch.int32, device=dev)
        _lib.call('crfconv_neighbor_maxpool_affine_forward', ptr(y), ptr(coef), ptr(table.idx32), table.K, table.m_tgt, co,
                  ptr(out), ptr(arg), st)
        ctx.table = table
        ctx.prm = (W, gamma, beta)
        ctx.save_for_backward(x, Wc, y, coef, arg)
        if fork:
            ctx.set_materialize_grads(False)
            return out, x_in
        return out

    @staticmethod
    def backward(ctx, g, g_alias=None):
        x, W, y, coef, arg = ctx.saved_tensors
        table = ctx.table
        m, ci = x.shape
        co = W.shape[0]
        dev = x.device
        g = torch.zeros((table.m_tgt, co), dtype=torch.float32, device=dev) if g is None else _f32c(g)
        st = stream_ptr()
        rev_ptr, rev_eid = table.reverse
        gA = torch.empty((m, co), dtype=torch.float32, device=dev)
        _lib.call('crfconv_neighbor_maxpool_backward', ptr(g), ptr(arg), ptr(rev_ptr), ptr(rev_eid), table.K, m, co, ptr(gA), st)
        dX = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        add = _f32c(g_alias) if (g_alias is not None and dX is not None) else None
        outs, dfr = _mlp_param_outs(ctx.prm, W, dev)       # (dW, dgamma, dbeta) targets; dfr: dW finished at the end of the pass
        dW, dgamma, dbeta = (o[0] for o in outs)
        nbytes = _lib.load().crfconv_mlp_backward_workspace(m, ci, co)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.call('crfconv_mlp_backward_add', ptr(gA), ptr(y), ptr(x), ptr(W), ptr(coef), 1.0, m, ci, co, ptr(add), ptr(dX),
                  ptr(None if dfr else dW), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, _mlp_ticket(dev), st)
        return dX, *_mlp_param_rets(ctx.prm, outs, dfr, ws, m, ci, co, coef), None, None, None, None, None, None


def mlp_block_pool(x, W, bn, table, fork=False):
    """neighbor_maxpool(BatchNorm_train(x W^T), table) as one node where the big-level fused block applies (x [m_src, Ci]
    rows of the table's source level), else None.  fork=True: returns (pooled, x_alias) -- see mlp_block."""
    if state.no_join or table.padded:
        return None
    m, ci = x.shape
    co = W.shape[0]
    if m != table.m_src or not (mlp_block_ok(x, W, None, bn, True) and not _mlp_small_ok(m, ci, co)):
        return None
    require_gpu(x, W)
    tick(bn)
    mom = 0.1 if bn.momentum is None else bn.momentum
    if fork and x.requires_grad and torch.is_grad_enabled() and not state.no_fork:
        return _MLPBlockPool.apply(x, W, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps, table, True)
    out = _MLPBlockPool.apply(x, W, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps, table, False)
    return (out, x) if fork else out




def _small_bwd_jobs(jobs, n, dev, st):
    """The BatchNorm(+LeakyReLU) backward + dX product of up to four coarse-level blocks: ONE launch whose product workgroups wait for
    its tile-sum workgroups (crfconv_mlp_small_backward_jobs_one_launch), or -- state.small_bwd_one_launch off, or after a wait inside
    a one-launch kernel has ever given up (check_gridsync) -- the two launches.  Bit-identical results."""
    if state.small_bwd_one_launch and not state.small_mlp_disabled:
        _lib.call('crfconv_mlp_small_backward_jobs_one_launch', ctypes.cast(jobs, ctypes.c_void_p), n, ptr(_ticket(dev)), ptr(gridsync_ws(dev)), st)
    else:
        _lib.call('crfconv_mlp_small_backward_jobs', ctypes.cast(jobs, ctypes.c_void_p), n, ptr(_ticket(dev)), st)


def _small_bwd(gA, y, coef, W, addend, slope, dgamma, dbeta, need_dx):
    """(gY, dX) of a coarse-level MLP block: BatchNorm(+LeakyReLU) backward and dX = gY W (+ addend) as TWO launches
    (crfconv_mlp_small_backward: tile sums, then the product with gY formed in its operand load), else -- no dX wanted, widths the
    fused form does not take -- crfconv_bn_backward followed by the plain product."""
    m, co = y.shape
    ci = W.shape[1]
    dev = y.device
    gY = torch.empty_like(y)
    lib = _lib.load()
    if need_dx and lib.crfconv_mlp_small_backward_supported(m, ci, co) == 1:
        dX = torch.empty((m, ci), dtype=torch.float32, device=dev)
        nbytes = lib.crfconv_mlp_small_backward_workspace(m, co)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        add = None if addend is None else addend.contiguous()
        jobs = (_lib.MlpBwdJob * 1)()
        jobs[0] = _lib.MlpBwdJob(gA.data_ptr(), y.data_ptr(), coef.data_ptr(), W.data_ptr(), None if add is None else add.data_ptr(), m, ci, co, 1,
                                 float(slope), gY.data_ptr(), dX.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), nbytes)
        _small_bwd_jobs(jobs, 1, dev, stream_ptr())
        return gY, dX
    nbytes = lib.crfconv_bn_workspace(m, co)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.call('crfconv_bn_backward', ptr(gA), ptr(y), ptr(coef), m, co, 1, float(slope), ptr(gY), ptr(dgamma), ptr(dbeta), ptr(ws),
              nbytes, stream_ptr())
    return gY, (_gemm(gY, W, addend=addend) if need_dx else None)


def _small_fwd(x, Wc, gamma, beta, run_mean, run_var, momentum, eps, slope, skip=None, join_slope=1.0):
    """(y, out, coef) of a coarse-level MLP block: the one-launch kernel of csrc/mlp_small.hip where its workgroups are co-resident
    (crfconv_mlp_small_supported), else -- the rows between that limit and the switch-over to the row-streaming forms -- the tiled
    product with statistic records in its epilogue, the coefficient launch and one apply pass (with the join's add + LeakyReLU)."""
    m, ci = x.shape
    co = Wc.shape[0]
    dev = x.device
    lib = _lib.load()
    y = torch.empty((m, co), dtype=torch.float32, device=dev)
    out = torch.empty_like(y)
    coef = torch.empty(4 * co, dtype=torch.float32, device=dev)
    g, b = _f32c(gamma), _f32c(beta)
    if not state.small_mlp_disabled and lib.crfconv_mlp_small_supported(m, ci, co) == 1:
        nbytes = lib.crfconv_mlp_small_workspace(m, co)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        sync = gridsync_ws(dev)
        if skip is None:
            _lib.call('crfconv_mlp_small_forward', ptr(x), ptr(Wc), m, ci, co, ptr(g), ptr(b), ptr(run_mean), ptr(run_var),
                      float(momentum), float(eps), float(slope), ptr(y), ptr(out), ptr(coef), ptr(ws), nbytes, ptr(sync),
                      sync.numel() * 4, stream_ptr())
        else:
            _lib.call('crfconv_mlp_small_forward_join', ptr(x), ptr(Wc), m, ci, co, ptr(g), ptr(b), ptr(run_mean), ptr(run_var),
                      float(momentum), float(eps), float(slope), ptr(skip), float(join_slope), ptr(y), ptr(out), ptr(coef), ptr(ws),
                      nbytes, ptr(sync), sync.numel() * 4, stream_ptr())
        return y, out, coef
    # product with the BatchNorm statistic records in its epilogue -> coefficients -> apply (+ skip, the join): three launches
    nrec = lib.crfconv_gemm_stat_records(m)
    rec = torch.empty((nrec, co, 4), dtype=torch.float32, device=dev)
    _lib.call('crfconv_gemm_stats', ptr(x), ptr(Wc), m, co, ci, ptr(y), ptr(rec), stream_ptr())
    _lib.call('crfconv_bn_apply_from_records', ptr(rec), nrec, ptr(y), m, co, ptr(g), ptr(b), ptr(run_mean), ptr(run_var),
              float(momentum), float(eps), ptr(skip), float(slope if skip is None else join_slope), ptr(coef), ptr(out), stream_ptr())
    return y, out, coef


class _MLPSmallJoin(torch.autograd.Function):
    """_MLPBlockJoin at the coarse levels: the one-launch Linear + BatchNorm kernel (csrc/mlp_small.hip) also adds the skip and
    applies the join's LeakyReLU to the tile it holds in registers (crfconv_mlp_small_forward_join)."""

    @staticmethod
    def forward(ctx, x, W, gamma, beta, run_mean, run_var, momentum, eps, skip, slope):
        x, Wc, skip = x.contiguous(), W.contiguous(), skip.contiguous()
        y, out, coef = _small_fwd(x, Wc, gamma, beta, run_mean, run_var, momentum, eps, 1.0, skip, slope)
        ctx.prm = (W, gamma, beta)
        ctx.save_for_backward(x, Wc, y, coef, out)
        ctx.slope = float(slope)
        ctx.params = (W, None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, W, y, coef, out = ctx.saved_tensors
        m, ci = x.shape
        co = W.shape[0]
        dev = x.device
        g = g.contiguous()
        st = stream_ptr()
        g1 = torch.empty_like(g)
        _lib.call('crfconv_add_lrelu_backward', ptr(g), ptr(out), out.numel(), ctx.slope, ptr(g1), st)
        outs = [_param_out(q, (co,), dev) for q in ctx.prm[1:]]      # (dgamma, dbeta) targets
        dgamma, dbeta = outs[0][0], outs[1][0]
        gY, dX = _small_bwd(g1, y, coef, W, None, 1.0, dgamma, dbeta, ctx.needs_input_grad[0])
        gskip = g1 if ctx.needs_input_grad[8] else None
        if _defer_ok(ctx.params):
            _defer_weight_grad(gY, x, ctx.params, False)
            return dX, None, *(_param_ret(q, o, k) for q, (o, k) in zip(ctx.prm[1:], outs)), None, None, None, None, gskip, None
        dW = torch.empty((co, ci), dtype=torch.float32, device=dev)
        nb = _lib.load().crfconv_linear_wgrad_workspace(m, co, ci)
        wsw = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.call('crfconv_linear_wgrad', ptr(gY), ptr(x), m, co, ci, ptr(dW), None, ptr(wsw), nb, st)
        return dX, dW, *(_param_ret(q, o, k) for q, (o, k) in zip(ctx.prm[1:], outs)), None, None, None, None, gskip, None


def mlp_block_join(x, W, bn, skip, slope):
    """lrelu(BatchNorm_train(x W^T) + skip, slope) as one node where the big-level fused block applies, else None (the
    caller then runs its own lin_out + add_lrelu)."""
    if state.no_join or skip.shape[:-1] != x.shape[:-1] or skip.shape[-1] != W.shape[0] or skip.dtype != torch.float32:
        return None
    m = x.numel() // x.shape[-1]
    ci, co = x.shape[-1], W.shape[0]
    if not mlp_block_ok(x, W, None, bn, True):
        return None
    require_gpu(x, W, skip)
    tick(bn)
    mom = 0.1 if bn.momentum is None else bn.momentum
    fn = _MLPSmallJoin if _mlp_small_ok(m, ci, co) else _MLPBlockJoin      # coarse levels: folded into the one-launch kernel
    out = fn.apply(x.reshape(-1, ci), W, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps,
                   skip.reshape(-1, co), slope)
    return out.reshape(x.shape[:-1] + (co,))


def _mlp_small_ok(m, ci, co):
    if m >= state.mfma_min_rows:
        return False
    lib = _lib.load()
    if not state.small_mlp_disabled and lib.crfconv_mlp_small_supported(m, ci, co) == 1:
        return True                                          # forward in one launch
    # past the one-launch kernel's co-residency limit (or after a barrier failure): the same autograd nodes, forward as product +
    # BatchNorm launches (_small_fwd), backward as always (_small_bwd)
    return ci % 4 == 0 and co % 4 == 0 and lib.crfconv_mlp_small_backward_supported(m, ci, co) == 1


class _MLPSmall(torch.autograd.Function):
    """_MLPBlock for the coarse levels (m <= 4096 rows): forward in ONE launch (csrc/mlp_small.hip: MFMA tile, statistic
    records, grid barrier, BatchNorm + LeakyReLU on the tile in registers).  fork: as _MLPBlock -- the alias' gradient is the
    addend of the dX product (the GEMM's beta = 1 epilogue)."""

    @staticmethod
    def forward(ctx, x_in, W, gamma, beta, run_mean, run_var, momentum, eps, slope, fork=False):
        x = x_in.contiguous()
        Wc = W.contiguous()
        y, out, coef = _small_fwd(x, Wc, gamma, beta, run_mean, run_var, momentum, eps, slope)
        ctx.prm = (W, gamma, beta)
        ctx.save_for_backward(x, Wc, y, coef)
        ctx.slope = float(slope)
        ctx.params = (W, None)
        if fork:
            ctx.set_materialize_grads(False)
            return out, x_in
        return out

    @staticmethod
    def backward(ctx, gA, g_alias=None):
        x, W, y, coef = ctx.saved_tensors
        m, ci = x.shape
        co = W.shape[0]
        dev = x.device
        gA = torch.zeros_like(y) if gA is None else gA.contiguous()
        outs = [_param_out(q, (co,), dev) for q in ctx.prm[1:]]      # (dgamma, dbeta) targets
        dgamma, dbeta = outs[0][0], outs[1][0]
        gY, dX = _small_bwd(gA, y, coef, W, None if g_alias is None else g_alias.reshape(m, ci), ctx.slope, dgamma, dbeta,
                            ctx.needs_input_grad[0])
        if _defer_ok(ctx.params):
            _defer_weight_grad(gY, x, ctx.params, False)
            return dX, None, *(_param_ret(q, o, k) for q, (o, k) in zip(ctx.prm[1:], outs)), None, None, None, None, None, None
        dW = torch.empty((co, ci), dtype=torch.float32, device=dev)          # same partials + reduction as the deferred form
        nb = _lib.load().crfconv_linear_wgrad_workspace(m, co, ci)
        wsw = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.call('crfconv_linear_wgrad', ptr(gY), ptr(x), m, co, ci, ptr(dW), None, ptr(wsw), nb, stream_ptr())
        return dX, dW, *(_param_ret(q, o, k) for q, (o, k) in zip(ctx.prm[1:], outs)), None, None, None, None, None, None




class _MLPSmallGroup(torch.autograd.Function):
    """n INDEPENDENT coarse-level MLP blocks (each _MLPSmall's arithmetic) as one node: the forward is ONE product launch with the
    BatchNorm statistic records of every block (crfconv_gemm_stats_jobs) and ONE coefficient + apply launch
    (crfconv_bn_apply_from_records_jobs), the backward TWO launches for all blocks (crfconv_mlp_small_backward_jobs) -- a coarse launch is
    a latency chain on a fraction of the chip, so blocks whose inputs are both ready run side by side for the price of the longer one:
    unary_nn[i] / pairwise_nn[i] of a CRF layer (models/continuous_crf_conv_big.py:56-60), shortcut / lin_in of a strided ResNet block
    (models/point_conv_big.py:79-88).  Per block: (x, W, gamma, beta, run_mean, run_var, momentum, eps, slope, fork); a block with
    fork returns (out, x_alias) as _MLPBlock does.  shared: blocks 0 and 1 read the SAME tensor -- their input gradients are summed by
    one library launch here and returned once (autograd would add them with a framework kernel)."""

    NARG = 10

    @staticmethod
    def forward(ctx, shared, *args):
        n = len(args) // _MLPSmallGroup.NARG
        jobs = [args[i * _MLPSmallGroup.NARG:(i + 1) * _MLPSmallGroup.NARG] for i in range(n)]
        lib = _lib.load()
        st = stream_ptr()
        keep, outs, prm, slopes, forks, tmp = [], [], [], [], [], []
        gs = (_lib.GemmStatsJob * n)()
        ba = (_lib.BnApplyJob * n)()
        for i, (x_in, W, gamma, beta, rm, rv, mom, eps, slope, fork) in enumerate(jobs):
            if x_in is None:                           # shared: block 1 reads block 0's input (handed over once: ONE consumer in the graph)
                x_in = jobs[0][0]
            x, Wc = x_in.contiguous(), W.contiguous()
            m, ci = x.shape
            co = Wc.shape[0]
            dev = x.device
            y = torch.empty((m, co), dtype=torch.float32, device=dev)
            out = torch.empty_like(y)
            coef = torch.empty(4 * co, dtype=torch.float32, device=dev)
            nrec = lib.crfconv_gemm_stat_records(m)
            rec = torch.empty((nrec, co, 4), dtype=torch.float32, device=dev)
            g, b = _f32c(gamma), _f32c(beta)
            gs[i] = _lib.GemmStatsJob(x.data_ptr(), Wc.data_ptr(), m, co, ci, y.data_ptr(), rec.data_ptr())
            ba[i] = _lib.BnApplyJob(rec.data_ptr(), nrec, y.data_ptr(), m, co, g.data_ptr(), b.data_ptr(),
                                    None if rm is None else rm.data_ptr(), None if rv is None else rv.data_ptr(), float(mom), float(eps),
                                    None, float(slope), coef.data_ptr(), out.data_ptr())
            keep += [x, Wc, y, coef]
            prm.append((W, gamma, beta))
            slopes.append(float(slope))
            forks.append(bool(fork))
            outs.append(out)
            if fork:
                outs.append(x_in)
            tmp.append((rec, g, b))                    # alive until the launches below are queued
        _lib.call('crfconv_gemm_stats_jobs', ctypes.cast(gs, ctypes.c_void_p), n, st)
        _lib.call('crfconv_bn_apply_from_records_jobs', ctypes.cast(ba, ctypes.c_void_p), n, st)
        del tmp
        ctx.n, ctx.prm, ctx.slopes, ctx.forks, ctx.shared = n, prm, slopes, forks, bool(shared)
        ctx.save_for_backward(*keep)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        n = ctx.n
        saved = ctx.saved_tensors
        lib = _lib.load()
        st = stream_ptr()
        jobs = (_lib.MlpBwdJob * n)()
        per, gi = [], 0
        for i in range(n):
            x, W, y, coef = saved[4 * i:4 * i + 4]
            m, ci = x.shape
            co = W.shape[0]
            dev = x.device
            gA = grads[gi]
            gi += 1
            g_alias = None
            if ctx.forks[i]:
                g_alias = grads[gi]
                gi += 1
            gA = torch.zeros_like(y) if gA is None else gA.contiguous()
            need_dx = ctx.needs_input_grad[1 + (0 if (ctx.shared and i == 1) else i) * _MLPSmallGroup.NARG]
            outs = [_param_out(q, (co,), dev) for q in ctx.prm[i][1:]]      # (dgamma, dbeta) targets
            gY = torch.empty_like(y)
            dX = torch.empty((m, ci), dtype=torch.float32, device=dev)
            nbytes = lib.crfconv_mlp_small_backward_workspace(m, co)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            add = None if g_alias is None else _f32c(g_alias).reshape(m, ci)
            jobs[i] = _lib.MlpBwdJob(gA.data_ptr(), y.data_ptr(), coef.data_ptr(), W.data_ptr(), None if add is None else add.data_ptr(), m, ci, co, 1,
                                     ctx.slopes[i], gY.data_ptr(), dX.data_ptr(), outs[0][0].data_ptr(), outs[1][0].data_ptr(), ws.data_ptr(), nbytes)
            per.append((x, W, gY, dX, outs, need_dx, (gA, add, ws)))
        _small_bwd_jobs(jobs, n, per[0][0].device, st)
        rets = [None]
        dxs = [p[3] if p[5] else None for p in per]
        if ctx.shared and dxs[0] is not None and dxs[1] is not None:
            tot = torch.empty_like(dxs[0])
            _lib.call('crfconv_add_lrelu', ptr(dxs[0]), ptr(dxs[1]), tot.numel(), 1.0, ptr(tot), st)
            dxs[0], dxs[1] = tot, None
        for i, (x, W, gY, dX, outs, need_dx, _) in enumerate(per):
            m, ci = x.shape
            co = W.shape[0]
            Wp, gp, bp = ctx.prm[i]
            if _defer_ok((Wp, None)):
                _defer_weight_grad(gY, x, (Wp, None), False)
                dW = None
            else:
                dW = torch.empty((co, ci), dtype=torch.float32, device=x.device)
                nb = lib.crfconv_linear_wgrad_workspace(m, co, ci)
                wsw = torch.empty(nb, dtype=torch.uint8, device=x.device)
                _lib.call('crfconv_linear_wgrad', ptr(gY), ptr(x), m, co, ci, ptr(dW), None, ptr(wsw), nb, st)
            rets += [dxs[i], dW, _param_ret(gp, outs[0][0], outs[0][1]), _param_ret(bp, outs[1][0], outs[1][1]), None, None, None, None, None, None]
        return tuple(rets)


def mlp_group(blocks, shared=False):
    """[(x [.., Ci], W, bn, slope, fork)] -> per block its output (or (out, x_alias) with fork), all blocks in ONE node -- or None when
    the group form does not apply to every block (training-mode coarse-level blocks: _mlp_small_ok rows, affine float32 BatchNorm with
    running statistics, widths the two-launch backward takes).  shared: blocks 0 and 1 read the same tensor."""
    if not (2 <= len(blocks) <= 4):
        return None
    lib = _lib.load()
    args, shapes = [], []
    for x, W, bn, slope, fork in blocks:
        ci, co = x.shape[-1], W.shape[0]
        m = x.numel() // max(ci, 1)
        if not (x.is_cuda and x.dtype == torch.float32 and W.dtype == torch.float32 and bn.affine and bn.running_mean is not None
                and m >= 1 and _mlp_small_ok(m, ci, co) and ci % 4 == 0 and co % 4 == 0
                and lib.crfconv_mlp_small_backward_supported(m, ci, co) == 1):
            return None
    if shared and (state.no_fork or blocks[0][0] is not blocks[1][0]):
        return None                                    # (tests: the un-forked graph runs the blocks one by one)
    for i, (x, W, bn, slope, fork) in enumerate(blocks):
        tick(bn)
        mom = 0.1 if bn.momentum is None else bn.momentum
        use_fork = bool(fork and x.requires_grad and torch.is_grad_enabled() and not state.no_fork)
        xa = None if (shared and i == 1) else x.reshape(-1, x.shape[-1])
        args += [xa, W, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps, float(slope), use_fork]
        shapes.append((x.shape, W.shape[0], fork, use_fork))
    res = list(_MLPSmallGroup.apply(bool(shared), *args))
    out = []
    for (xs, co, fork, use_fork), (x, _, _, _, _) in zip(shapes, blocks):
        o = res.pop(0).reshape(xs[:-1] + (co,))
        if fork:
            alias = res.pop(0).reshape(xs) if use_fork else x
            out.append((o, alias))
        else:
            out.append(o)
    return out


class _MLPBlockCat(torch.autograd.Function):
    """_MLPBlock on the column concatenation [xa | xb] without materialising it (the CRF layers'
    fusion_nn(cat[x, pairwise]), models/continuous_crf_conv_big.py:76): the MFMA kernels read the two operands through
    two pointers and the backward writes the two input gradients separately -- no torch.cat, no slice copies."""

    @staticmethod
    def forward(ctx, xa, xb, W, gamma, beta, run_mean, run_var, momentum, eps, slope):
        xa, xb, Wc = xa.contiguous(), xb.contiguous(), W.contiguous()
        m, split = xa.shape
        ci, co = split + xb.shape[1], Wc.shape[0]
        y = torch.empty((m, co), dtype=torch.float32, device=xa.device)
        nrec = _lib.load().crfconv_linear_forward_stat_records(m)
        rec = torch.empty((nrec, 4, co), dtype=torch.float32, device=xa.device)
        st = stream_ptr()
        _lib.call('crfconv_linear_forward_cat', ptr(xa), ptr(xb), split, ptr(Wc), None, m, ci, co, ptr(y), ptr(rec), st)
        coef = torch.empty(4 * co, dtype=torch.float32, device=xa.device)
        out = torch.empty_like(y)
        _lib.call('crfconv_bn_apply_from_records', ptr(rec), nrec, ptr(y), m, co, ptr(_f32c(gamma)), ptr(_f32c(beta)), ptr(run_mean),
                  ptr(run_var), float(momentum), float(eps), None, float(slope), ptr(coef), ptr(out), st)
        ctx.prm = (W, gamma, beta)
        ctx.save_for_backward(xa, xb, Wc, y, coef)
        ctx.slope = float(slope)
        return out

    @staticmethod
    def backward(ctx, gA):
        xa, xb, W, y, coef = ctx.saved_tensors
        m, split = xa.shape
        ci, co = split + xb.shape[1], W.shape[0]
        gA = gA.contiguous()
        dev = xa.device
        want_dx = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        dxa = torch.empty_like(xa) if want_dx else None
        dxb = torch.empty_like(xb) if want_dx else None
        outs, dfr = _mlp_param_outs(ctx.prm, W, dev)       # (dW, dgamma, dbeta) targets; dfr: dW finished at the end of the pass
        dW, dgamma, dbeta = (o[0] for o in outs)
        nbytes = _lib.load().crfconv_mlp_backward_workspace(m, ci, co)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.call('crfconv_mlp_backward_cat', ptr(gA), ptr(y), ptr(xa), ptr(xb), split, ptr(W), ptr(coef), ctx.slope, m, ci, co,
                  ptr(dxa), ptr(dxb), ptr(None if dfr else dW), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, _mlp_ticket(dev), stream_ptr())
        return (dxa if ctx.needs_input_grad[0] else None, dxb if ctx.needs_input_grad[1] else None,
                *_mlp_param_rets(ctx.prm, outs, dfr, ws, m, ci, co, coef),
                None, None, None, None, None)


class _Cat2(torch.autograd.Function):
    """torch.cat([xa, xb], -1) on [m, ca] / [m, cb] rows as one library launch; the backward hands back two CONTIGUOUS
    gradients from one pass (autograd's own backward returns strided slices, which every consumer then copies)."""

    @staticmethod
    def forward(ctx, xa, xb):
        xa, xb = xa.contiguous(), xb.contiguous()
        m, ca = xa.shape
        cb = xb.shape[1]
        out = torch.empty((m, ca + cb), dtype=torch.float32, device=xa.device)
        _lib.call('crfconv_cat2', ptr(xa), ptr(xb), m, ca, cb, ptr(out), stream_ptr())
        ctx.widths = (ca, cb)
        return out

    @staticmethod
    def backward(ctx, g):
        ca, cb = ctx.widths
        g = g.contiguous()
        m = g.shape[0]
        ga = torch.empty((m, ca), dtype=torch.float32, device=g.device)
        gb = torch.empty((m, cb), dtype=torch.float32, device=g.device)
        _lib.call('crfconv_split2', ptr(g), m, ca, cb, ptr(ga), ptr(gb), stream_ptr())
        return ga, gb


def cat2(xa, xb):
    """torch.cat([xa, xb], dim=-1) for two float32 CUDA tensors of equal leading shape (the fusion layers' input where the
    two-pointer Linear does not apply); other inputs go to torch.cat."""
    ca, cb = xa.shape[-1], xb.shape[-1]
    if not (xa.is_cuda and xb.is_cuda and xa.dtype == torch.float32 and xb.dtype == torch.float32
            and xa.shape[:-1] == xb.shape[:-1] and ca % 4 == 0 and cb % 4 == 0 and ca >= 4 and cb >= 4 and xa.numel() > 0):
        return torch.cat([xa, xb], dim=-1)
    out = _Cat2.apply(xa.reshape(-1, ca), xb.reshape(-1, cb))
    return out.reshape(xa.shape[:-1] + (ca + cb,))


def mlp_block_cat(xa, xb, W, bn, training, slope=1.0):
    """lrelu(BatchNorm(cat[xa, xb] W^T), slope): the two-pointer fused block where it applies (training, MFMA-sized rows,
    widths multiples of 4), else torch.cat + the one-operand path.  Returns None when the caller should run its own
    module path (so that non-fusable configurations keep their exact semantics)."""
    if xa.dim() != xb.dim() or xa.shape[:-1] != xb.shape[:-1]:
        return None
    ca, cb = xa.shape[-1], xb.shape[-1]
    m = xa.numel() // ca
    probe = xa.new_empty((1, ca + cb))
    if not (ca % 4 == 0 and cb % 4 == 0 and xb.dtype == torch.float32 and m >= state.mfma_min_rows
            and mlp_block_ok(probe.expand(m, ca + cb), W, None, bn, training)):
        return None
    require_gpu(xa, xb, W)
    tick(bn)
    mom = 0.1 if bn.momentum is None else bn.momentum
    out = _MLPBlockCat.apply(xa.reshape(-1, ca), xb.reshape(-1, cb), W, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom,
                             bn.eps, slope)
    return out.reshape(xa.shape[:-1] + (W.shape[0],))


def mlp_block_ok(x, W, bias, bn, training):
    """The fused block applies to the training-mode MLPs of the fine levels (MFMA-sized rows, affine BatchNorm, no bias)."""
    if not training or bias is not None or x.dtype != torch.float32 or not bn.affine:
        return False
    m = x.numel() // x.shape[-1]
    ci, co = x.shape[-1], W.shape[0]
    if bn.running_mean is None:
        return False
    if _mlp_small_ok(m, ci, co):
        return True
    return (_mfma_ok(m, ci, co) and co % 4 == 0
            and _lib.load().crfconv_mlp_backward_supported(m, ci, co) == 1)




def mlp_block(x, W, bn, slope=1.0, fork=False):
    """lrelu(BatchNorm_train(x W^T), slope) on [..., Ci] rows; `bn`: the torch.nn.BatchNorm1d with the parameters.
    fork=True returns (out, x_alias): hand x_alias to the OTHER consumer of x and its gradient is added inside this block's
    backward (see _MLPBlock)."""
    require_gpu(x, W)
    shape = x.shape
    tick(bn)
    mom = 0.1 if bn.momentum is None else bn.momentum
    x2 = x.reshape(-1, shape[-1])
    fn = _MLPSmall if _mlp_small_ok(x2.shape[0], shape[-1], W.shape[0]) else _MLPBlock
    if fork and x2.requires_grad and torch.is_grad_enabled() and not state.no_fork:
        out, alias = fn.apply(x2, W, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps, slope, True)
        return out.reshape(shape[:-1] + (W.shape[0],)), alias.reshape(shape)
    out = fn.apply(x2, W, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps, slope, False)
    out = out.reshape(shape[:-1] + (W.shape[0],))
    return (out, x) if fork else out


def run_lin_bn(seq, x):
    """Runs an ``nn.Sequential`` of the reference's sparse layers -- [Linear, BatchNorm1d(, LeakyReLU)] groups, e.g.
    models/continuous_crf_conv.py:24-38, models/point_conv.py:21-41 -- on the HIP operators: the Sequential only keeps
    the parameters (its indices ARE the checkpoint keys); each group becomes ops.mlp_block (fused MFMA Linear +
    BatchNorm + LeakyReLU with the two-pass backward) when it applies, else ops.linear + ops.bn_act."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        lin = mods[i]
        if not isinstance(lin, torch.nn.Linear):
            raise _lib.CrfConvError('run_lin_bn: expected a Linear at position %d, got %s' % (i, type(lin).__name__))
        bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.BatchNorm1d) else None
        j = i + 1 + (bn is not None)
        act = mods[j] if j < len(mods) and isinstance(mods[j], torch.nn.LeakyReLU) else None
        slope = act.negative_slope if act is not None else 1.0
        co = lin.out_features
        if bn is not None and co % 4 == 0 and co <= 1024 and bn.affine and x.dtype == torch.float32:
            if mlp_block_ok(x, lin.weight, lin.bias, bn, seq.training):
                x = mlp_block(x, lin.weight, bn, slope)
            else:
                records = None
                if seq.training:
                    x, records = linear(x, lin.weight, lin.bias, want_stats=True)
                else:
                    x = linear(x, lin.weight, lin.bias)
                x = bn_act(x, bn, seq.training, slope, records=records)
        else:
            x = linear(x, lin.weight, lin.bias)
            if bn is not None:
                if not bn.affine:
                    raise _lib.CrfConvError('run_lin_bn: affine BatchNorm only (csrc/bn.hip)')
                x = bn_act(x, bn, seq.training, slope)         # any width: the statistics + apply kernels of csrc/bn.hip
            elif act is not None:
                x = _LRelu.apply(x, slope)
        i = j + (act is not None)
    return x


# names of the sibling modules, imported LAST: every use is inside a function body, so import cycles between the families are harmless
from .rows import _LRelu  # noqa: E402
from .defer import _defer_ok, _defer_weight_grad, _mlp_param_outs, _mlp_param_rets, _param_out, _param_ret  # noqa: E402
from .dense import _gemm, _mfma_matmul, _mfma_ok, bn_act, linear, tick  # noqa: E402
