"""Continuous CRF mean field (dense and wide), its matrices and their riders, the discrete (label-space) CRF layer."""
import ctypes

import torch

from .. import _lib
from ..graph import NeighborTable, ptr, require_gpu, stream_ptr
from ._base import _f32c, _next_supported, _pad_channels, _ptr_array, _ticket, gridsync_ws, state

# ------------------------------------------------------------------------------ CRF mean field
def _table_is_local(table, m, k0, rows):
    """True when at least state.mf_block_min_locality of the table's entries point into the target's own block of `rows` consecutive
    rows (spatially sorted clouds).  Measured once per table and block size by one small launch whose count travels to pinned memory
    WITHOUT a host synchronisation: the FIRST forward over a table answers 'not local' (per-step launches -- equally fast)
    and leaves the measurement in flight; the second forward over the same table (a resident / static batch: the warm-up passes in front
    of a graph capture) reads it, waiting for it if it has to, and the verdict stays in table.cache -- a table refreshed in place keeps
    its first verdict.  A loop that builds a new table per step never waits.  Never measured inside a capture."""
    key = ('block_locality', rows)
    frac = table.cache.get(key)
    if frac is None:
        if torch.cuda.is_current_stream_capturing():
            return False
        pend = table.cache.get(('block_locality_pending', rows))
        if pend is None:
            from ..graph import _pinned_slot
            count = torch.zeros(1, dtype=torch.int64, device=table.idx32.device)
            _lib.call('crfconv_block_locality', ptr(table.idx32), m, table.K, k0, rows, ptr(count), stream_ptr())
            host = _pinned_slot(torch.int64)
            host.copy_(count, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            table.cache[('block_locality_pending', rows)] = (ev, host, count)
            return False
        ev, host, _ = pend
        ev.synchronize()
        frac = table.cache[key] = float(host[0]) / float(m * (table.K - k0))
        del table.cache[('block_locality_pending', rows)]
    return frac >= state.mf_block_min_locality


def _block_rows(table, m, H, k0, steps):
    """Rows per workgroup when this forward is to run as ONE launch with block-resident rows (csrc/crf_block.hip), else 0.
    `ops.state.mf_block`: 'off' never, 'on' whenever the shape is covered (tests), 'auto' (default) when in addition the launch can
    fill the chip (>= state.mf_block_min_rows rows) and the table is LOCAL: at least state.mf_block_min_locality of its entries
    point into the target's own block of consecutive rows -- true of spatially sorted clouds (the device collate emits Morton
    order), not of shuffled ones, whose neighbours the form would fetch past L1.  The fraction is measured once per table (one
    small launch and one host read, never inside a capture -- an unmeasured table takes the per-step kernels) and kept in
    table.cache: a table refreshed in place (MultiScaleData.load_, CollateGraph) keeps the verdict of its first batch."""
    mode = state.mf_block
    if mode == 'off' or table.padded or steps < 1:
        return 0
    rows = _lib.load().crfconv_meanfield_forward_block_rows(m, H, table.K, k0, steps)
    if rows <= 0 or mode == 'on':
        return rows
    if m < state.mf_block_min_rows:
        return 0
    return rows if _table_is_local(table, m, k0, rows) else 0


class _MeanField(torch.autograd.Function):
    """x_T of  x_0 = z,  x_t = z Q + (A x_{t-1}) P  with A = row-softmax(-|y_i - y_j|^2) over the
    table's columns k0..K-1  (models/continuous_crf_conv_big.py:49-54, 63-72)."""

    @staticmethod
    def forward(ctx, z, y, Q, P, table, k0, steps, late=False):
        require_gpu(z, y, Q, P)
        # `late`: the gradient box of crf_matrices_batched for this (Q, P) pair, or None.  With a box dP / dQ can wait for the end of the
        # backward pass like the matrices backward that consumes them -- and then they travel OUT OF BAND (the box), never as
        # autograd gradients: autograd would otherwise hold tensors that are only filled by the end-of-pass flush (a second
        # consumer of Q / P, a hook, retain_grad or anomaly mode would sum or inspect garbage)
        ctx.late = late if isinstance(late, dict) else None
        m, H = z.shape
        if m != table.m_tgt or y.shape[0] != m or table.m_src != m:
            raise _lib.CrfConvError('mean field: %d / %d rows for a table of %d targets over %d sources (the CRF graph '
                                    'lives on one point set)' % (m, y.shape[0], table.m_tgt, table.m_src))
        if table.padded and k0 != 0:
            raise _lib.CrfConvError('a padded (variable-degree) table has no self column: use k0 = 0')
        z, y, Q, P = _f32c(z), _f32c(y), _f32c(Q), _f32c(P)
        needs_grad = any(ctx.needs_input_grad[:4])
        if needs_grad and m * H * 4 >= 2 ** 31 and _lib.load().crfconv_meanfield_backward_supported(H, table.K, k0) == 1:
            # the backward kernels address row tables by 32-bit byte offsets: say so BEFORE any work is done, not after a forward that worked
            raise _lib.CrfConvError('mean field: a row table of %d x %d floats (>= 2 GiB) cannot be trained on (the backward addresses rows by '
                                    '32-bit byte offsets); split the batch' % (m, H))
        # inference with one step: the similarity weights are consumed inside the fused first kernel and never
        # re-read -- skip their 4K bytes/point store (a third of that kernel's traffic)
        keep_s = needs_grad or steps != 1 or k0 != 1 or table.K not in (16, 32) or m * H * 4 >= 2 ** 31
        s = torch.empty((m, table.K), dtype=torch.float32, device=z.device) if keep_s else None   # s[i*K + k]
        xs = torch.empty((max(steps, 1), m, H), dtype=torch.float32, device=z.device)
        if _block_rows(table, m, H, k0, steps) > 0:
            _lib.call('crfconv_meanfield_forward_block', ptr(z), ptr(y), ptr(table.idx32), ptr(table.idx16), table.n_tgt,
                      table.n_src, table.K, k0, m, H, ptr(Q), ptr(P), steps, ptr(s), ptr(xs), ptr(gridsync_ws(z.device)), stream_ptr())
        else:
            _lib.call('crfconv_meanfield_forward_u16', ptr(z), ptr(y), ptr(table.idx32), ptr(table.idx16), table.n_tgt,
                      table.n_src, table.K, k0, m, H, ptr(Q), ptr(P), steps, ptr(s), ptr(xs), stream_ptr())
        ctx.table, ctx.k0, ctx.steps = table, k0, steps
        if needs_grad:
            ctx.save_for_backward(z, y, Q, P, s, xs)
        if steps == 0:
            return z.clone()
        # a view of the saved iterates: an in-place edit downstream trips autograd's version check instead of
        # silently corrupting the backward pass, and the copy (m * H floats per layer) is saved
        return xs[steps - 1]

    @staticmethod
    def backward(ctx, gout):
        z, y, Q, P, s, xs = ctx.saved_tensors
        table, k0, T = ctx.table, ctx.k0, ctx.steps
        m, H = z.shape
        G = _f32c(gout)
        if T == 0:
            return G, torch.zeros_like(y), torch.zeros_like(Q), torch.zeros_like(P), None, None, None, None
        rev_ptr, rev_eid = table.reverse
        st = stream_ptr()
        lib = _lib.load()

        def skinny_tn(A, B, out):      # out = A^T B for [rows, H] operands: the MFMA row-reduction kernel
            rows = A.shape[0]
            wbytes = lib.crfconv_linear_wgrad_workspace(rows, H, H)
            wws = torch.empty(wbytes, dtype=torch.uint8, device=z.device)
            _lib.call('crfconv_linear_wgrad', ptr(A), ptr(B), rows, H, H, ptr(out), None, ptr(wws), wbytes, st)

        if lib.crfconv_meanfield_backward_supported(H, table.K, k0) == 1:
            # T + 1 launches (csrc/crf_bwd.hip): T - 1 reverse walks | one edge pass over all steps + softmax backward |
            # the last reverse walk with the dy scatter and the dP / dQ reduction riding along
            dev = z.device
            inside = lib.crfconv_meanfield_backward_param_grads_inside(H) == 1
            Gs = torch.empty((T, m, H), dtype=torch.float32, device=dev)        # entry 0 unused: G_T = gout
            dzq = torch.empty((m, H), dtype=torch.float32, device=dev)
            dz, dy_self, dy = (torch.empty_like(z) for _ in range(3))
            w = torch.empty_like(s)
            dP, dQ = torch.empty_like(P), torch.empty_like(Q)
            mts = sumG = None
            if not inside:
                mts = torch.empty((T, m, H), dtype=torch.float32, device=dev)
                sumG = torch.empty_like(z)
            wsb = lib.crfconv_meanfield_backward_workspace(m, H, table.K)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            _lib.call('crfconv_meanfield_backward', ptr(G), ptr(z), ptr(y), ptr(s), ptr(xs), ptr(table.idx32),
                      ptr(table.idx16), table.n_tgt, table.n_src, ptr(rev_ptr), ptr(rev_eid), table.K, k0, m, H, ptr(Q),
                      ptr(P), T, ptr(Gs), ptr(dzq), ptr(mts), ptr(sumG), ptr(dz), ptr(w), ptr(dy_self), ptr(dy), ptr(dP),
                      ptr(dQ), ptr(ws), wsb, ptr(_ticket(dev)), st)
            if not inside:                                  # (Gs[0] = G was written by the edge launch)
                if _late_box_ok(ctx.late):
                    # nothing on the chain reads dP / dQ, and their only consumer (the batched matrices backward) waits for the end
                    # of the pass too: partial passes and sums join the batched launches there; the buffers reach the matrices
                    # backward through the box, autograd gets None for Q and P
                    _defer_tn(mts.view(T * m, H), Gs.view(T * m, H), dP)
                    _defer_tn(z, sumG, dQ)
                    ctx.late['bufs'].append((dQ, dP))
                    return dz, dy, None, None, None, None, None, None
                else:
                    skinny_tn(mts.view(T * m, H), Gs.view(T * m, H), dP)
                    skinny_tn(z, sumG, dQ)
            return dz, dy, dQ, dP, None, None, None, None

        # generic shapes (any K <= 64 / k0, padded variable-degree tables): one edge + one scatter launch per step
        gm = torch.empty_like(z)
        ds = torch.empty_like(s)
        # G_t (gradient entering step t) and m_t for t = T..1 stacked row-wise: dP = sum_t m_t^T G_t and
        # sum_t G_t then take ONE row-reduction launch each instead of T accumulate passes
        Gs = torch.empty((T, m, H), dtype=torch.float32, device=z.device)
        mts = torch.empty((T, m, H), dtype=torch.float32, device=z.device)
        Gs[0].copy_(G)
        G0 = torch.empty_like(z)
        for i, t in enumerate(range(T, 0, -1)):
            xprev = xs[t - 2] if t >= 2 else z
            _lib.call('crfconv_meanfield_bwd_edge', ptr(Gs[i]), ptr(xprev), ptr(s), ptr(table.idx32), table.K, k0,
                      m, H, ptr(P), ptr(gm), ptr(ds), ptr(mts[i]), 0 if t == T else 1, st)
            _lib.call('crfconv_meanfield_bwd_scatter', ptr(gm), ptr(s), ptr(rev_ptr), ptr(rev_eid), table.K, k0,
                      m, H, None, ptr(Gs[i + 1] if i + 1 < T else G0), st)
        dP = torch.empty_like(P)
        skinny_tn(mts.view(T * m, H), Gs.view(T * m, H), dP)
        sumG = Gs.sum(0) if T > 1 else Gs[0]
        dz = _gemm(sumG, Q, addend=G0, nk=True)   # x_0 = z path + the z Q term of every step: G_0 + (sum_t G_t) Q^T
        dQ = torch.empty_like(Q)
        skinny_tn(z, sumG, dQ)
        w = torch.empty_like(s)
        dy_self = torch.empty_like(y)
        _lib.call('crfconv_similarity_bwd', ptr(ds), ptr(s), ptr(y), ptr(table.idx32), table.K, k0, m, H, ptr(w),
                  ptr(dy_self), st)
        dy = torch.empty_like(y)
        _lib.call('crfconv_similarity_bwd_scatter', ptr(w), ptr(y), ptr(dy_self), ptr(rev_ptr), ptr(rev_eid),
                  table.K, k0, m, H, ptr(dy), st)
        return dz, dy, dQ, dP, None, None, None, None


class _SpdInverse(torch.autograd.Function):
    """Q = M^-1 (M symmetric positive definite, H <= 64) on one workgroup; dM = -Q^T dQ Q^T."""

    @staticmethod
    def forward(ctx, M):
        require_gpu(M)
        Mc = _f32c(M)
        Q = torch.empty_like(Mc)
        _lib.call('crfconv_spd_inverse', ptr(Mc), Mc.shape[0], ptr(Q), stream_ptr())
        ctx.save_for_backward(Q)
        return Q

    @staticmethod
    def backward(ctx, gQ):
        (Q,) = ctx.saved_tensors
        return -_gemm(_gemm(Q, _f32c(gQ), nk=True), Q).t()     # Q^T gQ Q^T = (Q gQ^T Q)^T, both products on the library's tiled kernel


class _CrfMatrices(torch.autograd.Function):
    """c [H, H] -> Q = (I + c^T c)^-1, P = c^T c Q = I - Q: one workgroup forward, one backward (csrc/linear.hip)."""

    @staticmethod
    def forward(ctx, c):
        require_gpu(c)
        cc = _f32c(c)
        Q = torch.empty_like(cc)
        P = torch.empty_like(cc)
        _lib.call('crfconv_crf_matrices', ptr(cc), cc.shape[0], ptr(Q), ptr(P), stream_ptr())
        ctx.save_for_backward(cc, Q)
        return Q, P

    @staticmethod
    def backward(ctx, gQ, gP):
        cc, Q = ctx.saved_tensors
        gQ = None if gQ is None else _f32c(gQ)
        gP = None if gP is None else _f32c(gP)
        dc = torch.empty_like(cc)
        _lib.call('crfconv_crf_matrices_backward', ptr(cc), ptr(Q), ptr(gQ), ptr(gP), cc.shape[0], ptr(dc), stream_ptr())
        return dc


class _CrfMatricesWide(torch.autograd.Function):
    """c [H, H] -> Q = (I + c^T c)^-1, P = I - Q for 64 < H <= 512 (the wide stages of the sparse networks): c^T c on the row-reduction
    kernel, the inverse by crfconv_spd_inverse_wide, the backward dc = -c (T + T^T), T = Q (gQ - gP) Q, on the tiled product.  The
    only framework ops are element-wise ones on H x H parameter-sized matrices (identity, differences)."""

    @staticmethod
    def forward(ctx, c):
        require_gpu(c)
        cc = _f32c(c)
        H = cc.shape[0]
        M = _gemm_tn(cc, cc)
        M.diagonal().add_(1.0)
        Q = torch.empty_like(M)
        _lib.call('crfconv_spd_inverse_wide', ptr(M), H, ptr(Q), stream_ptr())
        P = Q.neg()
        P.diagonal().add_(1.0)
        ctx.save_for_backward(cc, Q)
        return Q, P

    @staticmethod
    def backward(ctx, gQ, gP):
        cc, Q = ctx.saved_tensors
        if gQ is None and gP is None:
            return None
        G = _f32c(gQ) if gP is None else (-_f32c(gP) if gQ is None else _f32c(gQ) - _f32c(gP))
        T = _gemm(_gemm(Q, G), Q)                          # d M = -Q^T G Q^T; Q is symmetric up to rounding
        dc = _gemm(cc, T, addend=_gemm(cc, T), nk=True)      # c T + c T^T
        return dc.neg_()


class _CrfMatricesBatched(torch.autograd.Function):
    """(Q_i, P_i) of several CRF layers from their factors c_i in ONE launch, and one launch for all dc_i: every
    layer's matrices depend on parameters only, so a network computes them together before its first layer
    (PointConvBig: four ~20 us single-workgroup launches each way become one)."""

    @staticmethod
    def forward(ctx, ride, *cs):
        require_gpu(*cs)
        ccs = [_f32c(c) for c in cs]
        Qs = [torch.empty_like(c) for c in ccs]
        Ps = [torch.empty_like(c) for c in ccs]
        Hs = (ctypes.c_int * len(ccs))(*[c.shape[0] for c in ccs])
        if ride:
            # the launch is QUEUED: the next PointConv statistics pass of a hosting width carries it (its workgroups ride along in
            # that launch: _take_riders), or flush_riders() -- called before anything reads Q / P -- issues it on its own
            flush_riders()
            _RIDERS['mats'] = (ccs, Hs, Qs, Ps)
        else:
            _lib.call('crfconv_crf_matrices_batched', _ptr_array(ccs), Hs, len(ccs), _ptr_array(Qs), _ptr_array(Ps), stream_ptr())
        ctx.save_for_backward(*ccs, *Qs)
        ctx.n = len(ccs)
        ctx.cparams = cs                           # the parameter objects themselves (late gradients are installed, not returned)
        ctx.set_materialize_grads(False)           # a layer whose dQ / dP arrive through its box gets None here, not zeros
        # one gradient box per layer: the mean-field nodes that defer dP / dQ to the end of the pass park their buffers here
        ctx.boxes = [{'bufs': [], 'cs': cs} for _ in ccs]
        out = []
        for Q, P in zip(Qs, Ps):
            out += [Q, P]
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        n = ctx.n
        ccs, Qs = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        gQ = [None if g is None else _f32c(g) for g in grads[0::2]]
        gP = [None if g is None else _f32c(g) for g in grads[1::2]]
        Hs = (ctypes.c_int * n)(*[c.shape[0] for c in ccs])
        boxes = ctx.boxes
        late = _late_box_ok(boxes[0])              # the SAME predicate the mean-field nodes used in this pass (_late_box_ok)
        if late:
            # the mean-field layers may have left dP / dQ to the batched sums at the end of the pass (_defer_tn): this launch goes
            # behind them (dc is a parameter gradient: nothing reads it before the pass is over) and, like every deferred weight
            # gradient, INSTALLS its results as .grad.  Gradients of a layer = what autograd delivered (consumers that did not
            # defer; None otherwise) + the buffers in its box (filled by the end-of-pass sums that run before this launch)
            outs = [_param_out(c, tuple(c.shape), c.device) for c in ctx.cparams]
            dcs = [o[0] for o in outs]
            cparams = ctx.cparams
            parked = [list(b['bufs']) for b in boxes]
            for b in boxes:
                b['bufs'].clear()                  # (a retained graph run again starts with empty boxes)

            def total(g, bufs):
                parts = ([g] if g is not None else []) + bufs
                if len(parts) <= 1:
                    return parts[0] if parts else None
                return torch.stack(parts).sum(0)   # several consumers of one (Q, P) pair: rare, a tiny eager sum

            def arguments():
                # (c, Q, gQ, gP, H, n, dc) of crfconv_crf_matrices_backward_batched, and what they point into: the end-of-pass flush
                # issues the launch -- on its own, or as the riders of the MLP blocks' weight-gradient launch (defer._flush_mlp_dw)
                gQs = [total(g, [b[0] for b in bufs]) for g, bufs in zip(gQ, parked)]
                gPs = [total(g, [b[1] for b in bufs]) for g, bufs in zip(gP, parked)]
                arrays = (_ptr_array(ccs), _ptr_array(Qs), _ptr_array(gQs), _ptr_array(gPs), Hs, n, _ptr_array(dcs))
                return arrays, (gQs, gPs)

            def install():
                for prm, (gr, direct) in zip(cparams, outs):
                    _install_grad(prm, gr, direct)
            _DEFER['late_mats'].append((arguments, install, (ccs, Qs, gQ, gP, dcs, parked)))
            _arm_flush()
            return (None,) * (n + 1)
        dcs = [torch.empty_like(c) for c in ccs]
        _lib.call('crfconv_crf_matrices_backward_batched', _ptr_array(ccs), _ptr_array(Qs), _ptr_array(gQ), _ptr_array(gP), Hs, n,
                  _ptr_array(dcs), stream_ptr())
        return (None,) + tuple(dcs)


_RIDERS = {'mats': None}      # a queued crf_matrices_batched launch (crf_matrices_batched(ride=True)) waiting for a launch to ride in


def flush_riders():
    """Issues a queued crf_matrices_batched launch on its own (no hosting launch came by): call before the first use of Q / P."""
    job, _RIDERS['mats'] = _RIDERS['mats'], None
    if job is not None:
        ccs, Hs, Qs, Ps = job
        _lib.call('crfconv_crf_matrices_batched', _ptr_array(ccs), Hs, len(ccs), _ptr_array(Qs), _ptr_array(Ps), stream_ptr())


def _take_riders(K, d):
    """The queued matrices launch for a PointConv statistics pass of this shape to carry (csrc/pointconv.hip:
    uvstats_hosting_kernel), or None."""
    if _RIDERS['mats'] is None or _lib.load().crfconv_pointconv_forward_uv_hosts(K, d) != 1:
        return None
    job, _RIDERS['mats'] = _RIDERS['mats'], None
    return job


def crf_matrices_batched(cs, ride=False):
    """[(Q, P)] for the compatibility factors `cs` (each [H, H], H <= 64) in one launch; None where a layer's H is wider
    (crf_meanfield then falls back to its own path).  ride: the launch is queued for the next PointConv statistics pass to carry
    (at most 8 layers; the caller runs flush_riders() before the first use of the matrices -- models/point_conv_big.py)."""
    idx = [i for i, c in enumerate(cs) if c.shape[0] <= _CRF_H[-1]]
    out = [None] * len(cs)
    for lo in range(0, len(idx), 8):
        part = idx[lo:lo + 8]
        res = _CrfMatricesBatched.apply(bool(ride) and len(idx) <= 8, *[cs[i] for i in part])
        node = res[0].grad_fn                      # the Function's ctx: carries one gradient box per layer
        for k, i in enumerate(part):
            if node is not None and hasattr(node, 'boxes'):
                res[2 * k]._crf_late = node.boxes[k]   # tells crf_meanfield where dQ / dP of this pair may be parked until the end of the pass
            out[i] = (res[2 * k], res[2 * k + 1])
    return out


def _late_box_ok(box):
    """True when dP / dQ of a mean-field layer may wait for the end of the backward pass: the pair comes from
    crf_matrices_batched (it has a gradient box), deferred weight gradients are on, and every factor of that batched node is a
    leaf parameter whose gradient the node's late launch can install.  Evaluated by the mean-field nodes AND by the matrices
    node inside one backward pass -- the same inputs, the same answer."""
    return (box is not None and _DEFER['on'] and all(_defer_ok((c, None)) for c in box['cs']))


_CRF_H = (4, 8, 16, 32, 64)
_CRF_WIDE_H = (128, 256)                    # one point per wavefront (crfconv_wide_*), H x H products as library GEMMs


class _MeanFieldWide(torch.autograd.Function):
    """The mean-field loop for H in {128, 256} (the 256- / 128-channel GCRFConv stages of the sparse networks,
    models/point_conv.py:318-339, on the coarsest point sets): the H x H tiles of csrc/crf.hip's kernels no longer fit
    LDS, so the graph work -- similarity soft-max, neighbour aggregation, the reverse-CSR scatters and the soft-max
    backward -- runs on the one-point-per-wavefront kernels (crfconv_wide_*), and the genuinely dense [m, H] x [H, H]
    products of each step run on this library's tiled MFMA product (csrc/gemm.hip) and row-reduction kernel (dP, dQ).  Same
    recurrence and gradients as _MeanField."""

    @staticmethod
    def forward(ctx, z, y, Q, P, table, k0, steps):
        require_gpu(z, y, Q, P)
        m, H = z.shape
        if m != table.m_tgt or y.shape[0] != m or table.m_src != m:
            raise _lib.CrfConvError('mean field: %d / %d rows for a table of %d targets over %d sources'
                                    % (m, y.shape[0], table.m_tgt, table.m_src))
        z, y, Q, P = _f32c(z), _f32c(y), _f32c(Q), _f32c(P)
        st = stream_ptr()
        K = table.K
        s = torch.empty((m, K), dtype=torch.float32, device=z.device)
        _lib.call('crfconv_wide_similarity', ptr(y), ptr(table.idx32), K, k0, m, H, ptr(s), st)
        zq = _gemm(z, Q)
        xs, msgs = [z], []
        for _ in range(steps):
            msg = torch.empty_like(z)
            _lib.call('crfconv_wide_aggregate', ptr(xs[-1]), ptr(s), ptr(table.idx32), K, k0, m, H, ptr(msg), st)
            msgs.append(msg)
            xs.append(_gemm(msg, P, addend=zq))
        ctx.table, ctx.k0, ctx.steps = table, k0, steps
        ctx.save_for_backward(z, y, Q, P, s, *xs[:-1], *msgs)
        return xs[-1] if steps > 0 else z.clone()

    @staticmethod
    def backward(ctx, gout):
        table, k0, T = ctx.table, ctx.k0, ctx.steps
        z, y, Q, P, s = ctx.saved_tensors[:5]
        xs, msgs = ctx.saved_tensors[5:5 + T], ctx.saved_tensors[5 + T:]
        m, H = z.shape
        K = table.K
        G = _f32c(gout)
        if T == 0:
            return G, torch.zeros_like(y), torch.zeros_like(Q), torch.zeros_like(P), None, None, None
        rev_ptr, rev_eid = table.reverse
        st = stream_ptr()
        ds = torch.empty_like(s)

        def add(a, b):                 # a + b in one library launch (the residual-join kernel at slope 1)
            out = torch.empty_like(a)
            _lib.call('crfconv_add_lrelu', ptr(a), ptr(b), a.numel(), 1.0, ptr(out), st)
            return out

        dP = sumG = None
        for t in range(T, 0, -1):
            gm = _gemm(G, P, nk=True)                      # G P^T
            _lib.call('crfconv_wide_bwd_edge', ptr(gm), ptr(xs[t - 1]), ptr(table.idx32), K, k0, m, H, ptr(ds),
                      0 if t == T else 1, st)
            dPt = _gemm_tn(msgs[t - 1], G)
            dP = dPt if dP is None else add(dP, dPt)
            sumG = G if sumG is None else add(sumG, G)
            Gprev = torch.empty_like(z)
            _lib.call('crfconv_wide_scatter', ptr(gm), ptr(s), ptr(rev_ptr), ptr(rev_eid), K, m, H, None, 0, ptr(Gprev), st)
            G = Gprev
        dz = _gemm(sumG, Q, addend=G, nk=True)             # G_0 + sum_t G_t Q^T
        dQ = _gemm_tn(z, sumG)
        w = torch.empty_like(s)
        dy_self, dy = torch.empty_like(y), torch.empty_like(y)
        _lib.call('crfconv_wide_similarity_bwd', ptr(ds), ptr(s), ptr(y), ptr(table.idx32), K, k0, m, H, ptr(w),
                  ptr(dy_self), st)
        _lib.call('crfconv_wide_scatter', ptr(y), ptr(w), ptr(rev_ptr), ptr(rev_eid), K, m, H, ptr(dy_self), 1, ptr(dy), st)
        return dz, dy, dQ, dP, None, None, None


def crf_meanfield(z, y, c, table, steps, k0=1, matrices=None):
    """z, y: [m, H] (flattened clouds);  c: [H, H] compatibility factor (C = c^T c).  `matrices` = (Q, P) of this c when
    the caller already has them (crf_matrices_batched)."""
    if _RIDERS['mats'] is not None:
        flush_riders()                  # queued matrices nobody carried: they must exist before this layer reads them
    H = z.shape[-1]
    if H > _CRF_WIDE_H[-1]:
        raise _lib.CrfConvError('mean field: H = %d exceeds the widest kernel (%d)' % (H, _CRF_WIDE_H[-1]))
    if H > _CRF_H[-1]:
        # Q = (I + c^T c)^-1 and P = I - Q (H x H, once per call: csrc/linear.hip spd_inverse_wide_kernel); zero-padded channels stay zero
        Hp = _next_supported(H, _CRF_WIDE_H)
        Q, P = _CrfMatricesWide.apply(c)
        if Hp != H:
            Q = torch.nn.functional.pad(Q, (0, Hp - H, 0, Hp - H))
            P = torch.nn.functional.pad(P, (0, Hp - H, 0, Hp - H))
        out = _MeanFieldWide.apply(_pad_channels(z, Hp), _pad_channels(y, Hp), Q, P, table, k0, steps)
        return out[:, :H] if Hp != H else out
    Q, P = matrices if matrices is not None else _CrfMatrices.apply(c)      # loop-invariant: once, not per step
    Hp = _next_supported(H, _CRF_H)
    if Hp != H:                                 # zero channels stay zero through every step
        Q = torch.nn.functional.pad(Q, (0, Hp - H, 0, Hp - H))
        P = torch.nn.functional.pad(P, (0, Hp - H, 0, Hp - H))
    out = _MeanField.apply(_pad_channels(z, Hp), _pad_channels(y, Hp), Q, P, table, k0, steps, getattr(Q, '_crf_late', None))
    return out[:, :H] if Hp != H else out


# ------------------------------------------------------------------------------ discrete (label-space) CRF layer
class _WeightedStep(torch.autograd.Function):
    """xout = z Q + (sum_k w_ik x_{j(i,k)}) P with GIVEN edge weights w [m, K] (models/discrete_crf_conv.py:58-59
    is this with Q = I, P = -C, z = -u).  Gradients to x, z, w, Q, P; scatter-free (reverse CSR)."""

    @staticmethod
    def forward(ctx, x, z, w, Q, P, table):
        require_gpu(x, z, w, Q, P)
        x, z, w, Q, P = _f32c(x), _f32c(z), _f32c(w), _f32c(Q), _f32c(P)
        m, H = x.shape
        out = torch.empty_like(x)
        _lib.call('crfconv_meanfield_step', ptr(x), ptr(z), ptr(w), ptr(table.idx32), table.K, 0, m, H, ptr(Q), ptr(P),
                  ptr(out), stream_ptr())
        ctx.table = table
        ctx.save_for_backward(x, z, w, Q, P)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, z, w, Q, P = ctx.saved_tensors
        table = ctx.table
        m, H = x.shape
        G = _f32c(gout)
        rev_ptr, rev_eid = table.reverse
        st = stream_ptr()
        gm, dw, mt, dx = torch.empty_like(x), torch.empty_like(w), torch.empty_like(x), torch.empty_like(x)
        _lib.call('crfconv_meanfield_bwd_edge', ptr(G), ptr(x), ptr(w), ptr(table.idx32), table.K, 0, m, H, ptr(P),
                  ptr(gm), ptr(dw), ptr(mt), 0, st)
        _lib.call('crfconv_meanfield_bwd_scatter', ptr(gm), ptr(w), ptr(rev_ptr), ptr(rev_eid), table.K, 0, m, H, None,
                  ptr(dx), st)
        dz = _gemm(G, Q, nk=True) if ctx.needs_input_grad[1] else None
        dQ = _gemm_tn(z, G) if ctx.needs_input_grad[3] else None
        dP = _gemm_tn(mt, G) if ctx.needs_input_grad[4] else None
        return dx, dz, dw, dQ, dP, None


class _KernelWeights(torch.autograd.Function):
    """w[i,k] = sum_g Wg[g] exp(-|fk[j,g,:] - fk[i,g,:]|^2) over the table's edges (discrete_crf_conv.py:49-54)."""

    @staticmethod
    def forward(ctx, fk, Wg, table, G, H):
        require_gpu(fk, Wg)
        fk, Wg = _f32c(fk), _f32c(Wg)
        m = fk.shape[0]
        w = torch.empty((m, table.K), dtype=torch.float32, device=fk.device)
        _lib.call('crfconv_kernel_weights_forward', ptr(fk), ptr(table.idx32), table.K, ptr(Wg), G, H, m, ptr(w),
                  stream_ptr())
        ctx.table, ctx.G, ctx.H = table, G, H
        ctx.save_for_backward(fk, Wg)
        return w

    @staticmethod
    def backward(ctx, gw):
        fk, Wg = ctx.saved_tensors
        table, G, H = ctx.table, ctx.G, ctx.H
        m = fk.shape[0]
        rev_ptr, rev_eid = table.reverse
        gw = _f32c(gw)
        scratch, dfk = torch.empty_like(fk), torch.empty_like(fk)
        nblk = _lib.load().crfconv_kernel_weights_partials(m) // 8
        part = torch.empty((nblk, 8), dtype=torch.float64, device=fk.device)
        _lib.call('crfconv_kernel_weights_backward', ptr(gw), ptr(fk), ptr(table.idx32), ptr(rev_ptr), ptr(rev_eid),
                  table.K, ptr(Wg), G, H, m, ptr(scratch), ptr(dfk), ptr(part), stream_ptr())
        return dfk, part.sum(0)[:G].to(torch.float32), None, None, None


def weighted_step(x, z, w, Q, P, table):
    return _WeightedStep.apply(x, z, w, Q, P, table)


def kernel_weights(fk, Wg, table, G, H):
    return _KernelWeights.apply(fk, Wg, table, G, H)


def discrete_meanfield(p, u, w, C, table, steps):
    """q_0 = p;  q <- softmax(-u - (sum_e w_e q_j) C)  `steps` times (models/discrete_crf_conv.py:56-61); label
    dimension padded to a kernel width, the soft-max taken over the real labels only."""
    L = p.shape[1]
    Hp = _next_supported(L, _CRF_H)
    eye = torch.eye(Hp, dtype=torch.float32, device=p.device)
    negC = torch.nn.functional.pad(-C, (0, Hp - L, 0, Hp - L))
    z = _pad_channels(-u, Hp)
    q = p
    for _ in range(steps):
        x = weighted_step(_pad_channels(q, Hp), z, w, eye, negC, table)
        q = torch.softmax(x[:, :L], dim=-1)
    return q


# names of the sibling modules, imported LAST: every use is inside a function body, so import cycles between the families are harmless
from .defer import _DEFER, _arm_flush, _defer_ok, _defer_tn, _install_grad, _param_out  # noqa: E402
from .dense import _gemm, _gemm_tn  # noqa: E402
