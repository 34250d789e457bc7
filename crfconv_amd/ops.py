"""torch.autograd bindings of the HIP kernels (through the C ABI; see include/crfconv_amd.h).

Every op here runs hand-written gfx950 kernels; torch supplies device memory, the stream and the
tiny dense algebra on H x H / d x d parameter matrices.
"""
import ctypes

import torch

from . import _lib
from .graph import NeighborTable, ptr, require_gpu, stream_ptr

BN_EPS = 1e-5


def _f32c(t):
    if t.dtype is torch.float32 and t.is_contiguous():
        return t.detach()
    return t.detach().to(torch.float32).contiguous()


def _pad_channels(t, C):
    """Zero-pad the last dim to C (kernels take power-of-two channel counts; zeros are exact)."""
    if t.shape[-1] == C:
        return t
    return torch.nn.functional.pad(t, (0, C - t.shape[-1]))


def _next_supported(c, choices):
    for v in choices:
        if c <= v:
            return v
    raise _lib.CrfConvError('channel count %d exceeds the largest supported (%d)' % (c, choices[-1]))


# ------------------------------------------------------------------------------ CRF mean field
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


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


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

            def launch():
                gQs = [total(g, [b[0] for b in bufs]) for g, bufs in zip(gQ, parked)]
                gPs = [total(g, [b[1] for b in bufs]) for g, bufs in zip(gP, parked)]
                _lib.call('crfconv_crf_matrices_backward_batched', _ptr_array(ccs), _ptr_array(Qs), _ptr_array(gQs), _ptr_array(gPs), Hs, n,
                          _ptr_array(dcs), stream_ptr())
                for prm, (gr, direct) in zip(cparams, outs):
                    _install_grad(prm, gr, direct)
            _DEFER['late_calls'].append((launch, (ccs, Qs, gQ, gP, dcs, parked)))
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


_TICKETS = {}


def _stream_key(device):
    """(device index, handle of the stream the caller launches on): the inter-workgroup scratch words below are per stream,
    so that launches on two streams of one device (a second training stream, evaluation beside training) never share
    barrier / ticket counts.  CAPTURED launches of a device all use ONE buffer (created by the eager warm-up pass, whatever
    stream the capture later runs on): hipGraphs that contain mean-field backward or one-launch MLP kernels must therefore be
    replayed one after the other (the loops of this package do) -- replaying two of them CONCURRENTLY on different streams is
    unsupported, their barrier and ticket counts would mix."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if torch.cuda.is_current_stream_capturing():
        return idx, 'capture'                      # captured launches: one buffer per device, created by the eager warm-up pass
    return idx, int(torch.cuda.current_stream(idx).cuda_stream)


def _stream_buf(table, device, make):
    """table[(device, stream)] (see _stream_key), created on first use; the eager pass that creates a stream's buffer also
    creates the device's capture buffer, so that a capture never allocates."""
    key = _stream_key(device)
    buf = table.get(key)
    if buf is None:
        buf = table[key] = make()
        if key[1] != 'capture' and (key[0], 'capture') not in table:
            table[(key[0], 'capture')] = make()
    return buf


def _ticket(device):
    """Zero words for the "last workgroup finishes" reductions (left zero by the kernels), per (device, stream)."""
    return _stream_buf(_TICKETS, device, lambda: torch.zeros(_lib.load().crfconv_ticket_bytes() // 4, dtype=torch.int32, device=device))




def _pc_ticket(device):
    return ptr(_ticket(device))




def _mlp_ticket(device):
    return ptr(_ticket(device))


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


# ------------------------------------------------------------------------------ deferred weight gradients
# Every Linear's dW = G^T X ends in a small "sum the row-slice partials" launch; PointConvBig has 74 of them per
# backward pass, each far below the cost of launching it.  Inside ``with deferred_weight_grads():`` the MFMA kernel
# only writes its partials, and ONE batched launch at the end of the backward pass (an autograd engine callback,
# like DDP's) finishes all of them and installs / accumulates ``.grad`` of the weight and bias parameters directly.
# Opt-in because it bypasses autograd for those leaves: ``torch.autograd.grad(loss, weight)`` sees nothing, and
# gradient hooks on the weights do not fire.  ``loss.backward()`` + ``param.grad`` behave as usual.
_DEFER = {'on': False, 'jobs': [], 'partials': [], 'red64': [], 'pc_wide': [], 'tn': [], 'late_calls': [], 'folds': [], 'mlpdw': [], 'armed': False, 'claimed': set()}


class deferred_weight_grads:
    """``sink``: optional callable  parameter -> preallocated gradient tensor (or None), e.g.
    ``FlatGradAllReduce.view_of``: the batched reduction then writes a weight gradient straight into the caller's flat
    bucket slice and installs that slice as ``.grad`` (no copy at pack time); parameters whose ``.grad`` already exists
    accumulate through a temporary as before."""

    def __init__(self, enabled=True, sink=None):
        self.enabled = enabled
        self.sink = sink

    def __enter__(self):
        self.prev = _DEFER['on']
        self.prev_sink = _DEFER.get('sink')
        _DEFER['on'] = bool(self.enabled)
        _DEFER['sink'] = self.sink
        if not self.prev:                 # outermost context: nothing of an earlier (failed) backward may linger
            _DEFER['jobs'], _DEFER['folds'], _DEFER['mlpdw'], _DEFER['armed'], _DEFER['claimed'] = [], [], [], False, set()
            for k in ('partials', 'red64', 'pc_wide', 'tn', 'late_calls'):
                _DEFER[k] = []
        return self

    def __exit__(self, exc_type, *exc):
        _DEFER['on'] = self.prev
        _DEFER['sink'] = self.prev_sink
        if exc_type is not None and not self.prev:
            # the backward raised after arming the engine callback: drop its queued partials, or every later backward
            # would find 'armed' set, never queue the callback again and silently lose all Linear weight gradients
            _DEFER['jobs'], _DEFER['folds'], _DEFER['mlpdw'], _DEFER['armed'], _DEFER['claimed'] = [], [], [], False, set()
            for k in ('partials', 'red64', 'pc_wide', 'tn', 'late_calls'):
                _DEFER[k] = []
        return False


def _defer_ok(params):
    W, b = params
    return (_DEFER['on'] and isinstance(W, torch.nn.Parameter) and W.is_leaf and W.requires_grad
            and (b is None or (isinstance(b, torch.nn.Parameter) and b.is_leaf)))


def _param_out(prm, shape, dev):
    """Where a kernel writes the gradient of parameter `prm`: inside ``deferred_weight_grads(sink=...)`` the caller's own
    storage for it (a flat-bucket slice), if `prm` is a leaf whose ``.grad`` is still unset -- then no copy at pack time --,
    else a new tensor that goes back through autograd.  Returns (tensor, direct)."""
    sink = _DEFER.get('sink') if _DEFER['on'] else None
    if (sink is not None and isinstance(prm, torch.nn.Parameter) and prm.is_leaf and prm.requires_grad and prm.grad is None
            and prm.dtype == torch.float32 and id(prm) not in _DEFER['claimed']):
        dst = sink(prm)
        if dst is not None and dst.dtype == torch.float32 and dst.is_contiguous() and dst.numel() == prm.numel():
            _DEFER['claimed'].add(id(prm))     # a second use of the parameter in this backward (shared weights) sums into it
            return dst.view(shape), True
    return torch.empty(shape, dtype=torch.float32, device=dev), False


def _param_ret(prm, buf, direct):
    """The value a backward returns for `prm`: None once the gradient sits in the caller's storage (installed as ``.grad``)."""
    if direct:
        prm.grad = buf.view_as(prm)
        return None
    return buf


_WGRAD_BATCH_ROWS = 65536          # partial passes of layers up to this many rows wait for the batched launch (0 / 200 000 rows measured no better)


def _defer_weight_grad(g, x, params, has_bias):
    m, Co = g.shape
    Ci = x.shape[1]
    lib = _lib.load()
    nbytes = lib.crfconv_linear_wgrad_workspace(m, Co, Ci)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
    want_b = bool(has_bias and params[1].requires_grad)
    if m <= _WGRAD_BATCH_ROWS:
        # the PARTIAL pass waits too: nothing on the backward chain reads it, and the coarse levels' passes are ~9 us launches of a
        # few workgroups each -- all of them go out in one launch per tile class at the end (crfconv_linear_wgrad_partial_jobs);
        # g and x stay alive until then (small: that is the point)
        g, x = g.contiguous(), x.contiguous()
        _DEFER['partials'].append((_lib.WgradJob(g.data_ptr(), x.data_ptr(), m, Co, Ci, 1 if want_b else 0, ws.data_ptr(), nbytes), (g, x)))
        nblk = lib.crfconv_linear_wgrad_nblk(m, Co, Ci)
    else:
        nb = ctypes.c_int(0)
        _lib.call('crfconv_linear_wgrad_partial', ptr(g), ptr(x), m, Co, Ci, 1 if want_b else 0, ptr(ws), nbytes,
                  ctypes.byref(nb), stream_ptr())
        nblk = nb.value
    _DEFER['jobs'].append((params[0], params[1] if want_b else None, ws, nblk, Co, Ci))
    _arm_flush()


def _defer_tn(A, B, out):
    """out [Ca, Cb] = A^T B ([m, Ca] / [m, Cb] rows) finished at the end of the backward pass: the partial pass joins
    crfconv_linear_wgrad_partial_jobs, the sum crfconv_reduce_jobs.  `out` is handed on now and filled then."""
    m, ca = A.shape
    cb = B.shape[1]
    lib = _lib.load()
    nbytes = lib.crfconv_linear_wgrad_workspace(m, ca, cb)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=A.device)
    A, B = A.contiguous(), B.contiguous()
    _DEFER['partials'].append((_lib.WgradJob(A.data_ptr(), B.data_ptr(), m, ca, cb, 0, ws.data_ptr(), nbytes), (A, B)))
    _DEFER['tn'].append((ws, lib.crfconv_linear_wgrad_nblk(m, ca, cb), ca * cb, out))
    _arm_flush()


def _arm_flush():
    if not _DEFER['armed']:
        _DEFER['armed'] = True
        torch.autograd.Variable._execution_engine.queue_callback(_flush_weight_grads)


def _defer_fold1_bwd(job, keep, installs):
    """Queues one PointConv layer's fold1_bwd (a _lib.Fold1BwdJob; `keep`: the tensors its pointers refer to; `installs`:
    (parameter, gradient tensor, direct) triples) for the single batched launch at the end of the backward pass."""
    _DEFER['folds'].append((job, keep, installs))
    _arm_flush()


def _mlp_dw_ret(deferred, prm, dW, direct, ws, m, ci, co, coef):
    """The weight gradient a fused MLP block's backward returns.  `deferred` (the C call was given dW = NULL): queue the slab
    reduction -- workspace, coefficients and target stay alive until then -- for the ONE launch that finishes the weight
    gradients of all blocks at the end of the backward pass (crfconv_mlp_dw_jobs), and return None."""
    if not deferred:
        return _param_ret(prm, dW, direct)
    _DEFER['mlpdw'].append((_lib.MlpDwJob(ws.data_ptr(), coef.data_ptr(), dW.data_ptr(), m, ci, co), (ws, coef), (prm, dW, direct)))
    _arm_flush()
    return None


def _mlp_param_outs(prm, W, dev):
    """Targets of (dW, dgamma, dbeta) of one fused MLP block -- [(tensor, direct)] * 3, see _param_out -- and whether the dW slab
    reduction is left to the batched launch at the end of the backward pass (then the C call gets dW = NULL)."""
    co = W.shape[0]
    outs = [_param_out(prm[0], tuple(W.shape), dev), _param_out(prm[1], (co,), dev), _param_out(prm[2], (co,), dev)]
    return outs, _defer_ok((prm[0], None))


def _mlp_param_rets(prm, outs, deferred, ws, m, ci, co, coef):
    """What the block's backward returns for (W, gamma, beta)."""
    (dW, kW), (dgamma, kg), (dbeta, kb) = outs
    return (_mlp_dw_ret(deferred, prm[0], dW, kW, ws, m, ci, co, coef), _param_ret(prm[1], dgamma, kg),
            _param_ret(prm[2], dbeta, kb))


def _install_grad(prm, gr, direct):
    """Gradient `gr` of one use of `prm` becomes / joins ``prm.grad``.  direct: `gr` IS the caller's storage for the
    parameter (a flat-bucket slice) -- it stays the ``.grad`` tensor and earlier contributions are added INTO it, so that
    views held by the caller stay valid; a parameter used twice in one backward (shared weights) sums."""
    gr = gr.view_as(prm)
    if prm.grad is None:
        prm.grad = gr
    elif direct:
        if prm.grad.data_ptr() != gr.data_ptr():
            gr.add_(prm.grad)
            prm.grad = gr
    else:
        prm.grad.add_(gr)


def _flush_mlp_dw():
    jobs, _DEFER['mlpdw'] = _DEFER['mlpdw'], []
    if not jobs:
        return
    table = (_lib.MlpDwJob * len(jobs))(*[j[0] for j in jobs])
    _lib.call('crfconv_mlp_dw_jobs', ctypes.cast(table, ctypes.c_void_p), len(jobs), stream_ptr())
    for _, _, (prm, gr, direct) in jobs:
        _install_grad(prm, gr, direct)


def _defer_reduce64(partial_ptr, is_float, nblk, nslots, out, keep):
    """Queues out[slot] = sum_b partial[b][slot] (float64; `keep`: the tensors the pointers refer to) for the one
    crfconv_reduce_jobs_f64 launch in front of the batched fold at the end of the backward pass."""
    _DEFER['red64'].append((_lib.Reduce64Job(partial_ptr, 1 if is_float else 0, int(nblk), int(nslots), out.data_ptr()), (out,) + tuple(keep)))
    _arm_flush()


def _take_red64():
    """The queued float64 sums (the PointConv layers' parameter-gradient slabs) as a job array for the end-of-pass sum launch."""
    red, _DEFER['red64'] = _DEFER.get('red64', []), []
    if not red:
        return None, 0, None
    return (_lib.Reduce64Job * len(red))(*[j for j, _ in red]), len(red), red


def _flush_fold1_bwd():
    """The batched BatchNorm-1 fold backward of all PointConv layers: AFTER the float64 sums it reads."""
    folds, _DEFER['folds'] = _DEFER['folds'], []
    if not folds:
        return
    table = (_lib.Fold1BwdJob * len(folds))(*[f[0] for f in folds])
    _lib.call('crfconv_pointconv_fold1_bwd_batched', ctypes.cast(table, ctypes.c_void_p), len(folds), stream_ptr())
    for _, _, installs in folds:
        for prm, gr, direct in installs:
            _install_grad(prm, gr, direct)


def _flush_pc_wide():
    """The parameter pass of all wide PointConv layers of this backward pass: dumps (one launch per width), the g_h2^T h1 partials
    (queued with every other weight-gradient partial pass), g_h1 = g_h2 W2 for all layers in one launch, the dA1 | db1 slab passes
    (one launch per width), their sums queued for the float64 reduce launch."""
    wide, _DEFER['pc_wide'] = _DEFER.get('pc_wide', []), []
    if not wide:
        return
    st = stream_ptr()
    lib = _lib.load()
    # d = 32 / 64 with K = 16: the whole pass on the matrix pipe, no per-edge tensor (csrc/pointconv_wide.hip) -- one launch per width;
    # its per-workgroup slabs join the batched sums below (dW2: crfconv_reduce_jobs, dA1 | db1: crfconv_reduce_jobs_f64)
    mfma = [w for w in wide if lib.crfconv_pointconv_wide_params_supported(w['m_tgt'], w['K'], w['d']) == 1]
    if mfma:
        wide = [w for w in wide if not any(w is v for v in mfma)]
        jobs, keep_m = [], []
        for w in mfma:
            d, dev = w['d'], w['x'].device
            nb = int(lib.crfconv_pointconv_wide_params_nblk(w['m_tgt'], d))
            pw = torch.empty((nb, d * d), dtype=torch.float32, device=dev)
            pa = torch.empty((nb, 4 * d), dtype=torch.float64, device=dev)
            c = w['coef']
            jobs.append(_lib.PcWideJob(w['x'].data_ptr(), w['g'].data_ptr(), w['pos_src'].data_ptr(), w['pos_tgt'].data_ptr(), w['idx'].data_ptr(),
                                       w['K'], w['m_tgt'], d, w['A1'].data_ptr(), w['b1'].data_ptr(), w['W2'].data_ptr(), float(w['slope']),
                                       c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), pw.data_ptr(), pa.data_ptr()))
            keep_m.append((w, pw, pa, nb))
        arr = (_lib.PcWideJob * len(jobs))(*jobs)
        _lib.call('crfconv_pointconv_wide_params_jobs', ctypes.cast(arr, ctypes.c_void_p), len(jobs), st)
        for w, pw, pa, nb in keep_m:
            d = w['d']
            _DEFER['jobs'].append((w['pW2'], None, pw, nb, d, d))          # dW2 = sum of the [d, d] slabs, installed with every other weight gradient
            _defer_reduce64(pa.data_ptr(), False, nb, 4 * d, w['dA1b1'], (pa, w))
        if not wide:
            return
    dumps, gemms, a1s, keep = [], [], [], []
    for w in wide:
        d, E, dev = w['d'], w['m_tgt'] * w['K'], w['x'].device
        h1 = torch.empty((E, d), dtype=torch.float32, device=dev)
        gh2 = torch.empty((E, d), dtype=torch.float32, device=dev)
        rel = torch.empty((E, 3), dtype=torch.float32, device=dev)
        gw = torch.empty((E, d), dtype=torch.float32, device=dev)
        abytes = lib.crfconv_pointconv_bwd_a1_workspace(E, d)
        aws = torch.empty(abytes, dtype=torch.uint8, device=dev)
        c = w['coef']
        dumps.append(_lib.PcDumpJob(w['x'].data_ptr(), w['g'].data_ptr(), w['pos_src'].data_ptr(), w['pos_tgt'].data_ptr(), w['idx'].data_ptr(),
                                    w['K'], w['m_tgt'], d, w['A1'].data_ptr(), w['b1'].data_ptr(), w['W2'].data_ptr(), float(w['slope']),
                                    c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), h1.data_ptr(), gh2.data_ptr(), rel.data_ptr()))
        gemms.append(_lib.GemmJob(gh2.data_ptr(), w['W2'].data_ptr(), gw.data_ptr(), E, d, d))
        a1s.append(_lib.PcA1Job(gw.data_ptr(), h1.data_ptr(), rel.data_ptr(), E, d, float(w['slope']), aws.data_ptr(), abytes))
        keep.append((w, h1, gh2, rel, gw, aws))
    arr = (_lib.PcDumpJob * len(dumps))(*dumps)
    _lib.call('crfconv_pointconv_bwd_dump_jobs', ctypes.cast(arr, ctypes.c_void_p), len(dumps), st)
    for w, h1, gh2, rel, gw, aws in keep:
        _defer_weight_grad(gh2, h1, (w['pW2'], None), False)          # (late mode implies the parameter is deferrable)
    arr = (_lib.GemmJob * len(gemms))(*gemms)
    _lib.call('crfconv_gemm_jobs', ctypes.cast(arr, ctypes.c_void_p), len(gemms), st)
    arr = (_lib.PcA1Job * len(a1s))(*a1s)
    _lib.call('crfconv_pointconv_bwd_a1_jobs', ctypes.cast(arr, ctypes.c_void_p), len(a1s), st)
    for w, h1, gh2, rel, gw, aws in keep:
        d, E = w['d'], w['m_tgt'] * w['K']
        _defer_reduce64((aws.data_ptr() + 255) & ~255, False, lib.crfconv_pointconv_bwd_a1_nblk(E, d), 4 * d, w['dA1b1'], (aws, gw, h1, rel))


def _flush_weight_grads():
    _flush_pc_wide()                       # first: it queues weight-gradient partials and float64 sums of its own
    jobs, _DEFER['jobs'], _DEFER['armed'] = _DEFER['jobs'], [], False
    partials, _DEFER['partials'] = _DEFER.get('partials', []), []
    if partials:
        # longest jobs first: their workgroups start first (the backward queues the fine levels -- the long jobs -- last)
        partials.sort(key=lambda e: -(e[0].M * e[0].Co * e[0].Ci))
        arr = (_lib.WgradJob * len(partials))(*[j for j, _ in partials])
        _lib.call('crfconv_linear_wgrad_partial_jobs', ctypes.cast(arr, ctypes.c_void_p), len(partials), stream_ptr())
    arr64, n64, keep64 = _take_red64()
    tns, _DEFER['tn'] = _DEFER.get('tn', []), []
    late_calls, _DEFER['late_calls'] = _DEFER.get('late_calls', []), []
    st = stream_ptr()
    if not jobs and not tns:
        if n64:
            _lib.call('crfconv_reduce_jobs_f64', ctypes.cast(arr64, ctypes.c_void_p), n64, st)
        _flush_fold1_bwd()
        _flush_mlp_dw()
        for fn, _ in late_calls:
            fn()
        _DEFER['claimed'] = set()
        return
    dev = (jobs[0][2] if jobs else tns[0][0]).device
    sink = _DEFER.get('sink')

    def direct(prm):                       # the caller's own gradient storage for this parameter, if it can be used as is
        if sink is None or prm.grad is not None or id(prm) in _DEFER['claimed']:
            return None                    # (claimed: an earlier job of this pass already writes there -- this one is added)
        dst = sink(prm)
        if dst is None or dst.dtype != torch.float32 or not dst.is_contiguous() or dst.numel() != prm.numel():
            return None
        _DEFER['claimed'].add(id(prm))
        return dst
    targets = [(direct(W), direct(b) if b is not None else None) for W, b, _, _, _, _ in jobs]
    total = sum((Co * Ci if tw is None else 0) + (Co if (b is not None and tb is None) else 0)
                for (_, b, _, _, Co, Ci), (tw, tb) in zip(jobs, targets))
    flat = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
    table = (_lib.ReduceJob * (2 * len(jobs) + len(tns)))()
    installs, n, o = [], 0, 0
    for ws_t, nblk_t, nslots_t, out_t in tns:          # plain A^T B sums (the CRF layers' dP / dQ): no parameter to install
        table[n] = _lib.ReduceJob(ws_t.data_ptr(), out_t.data_ptr(), nblk_t, nslots_t)
        n += 1
    for (W, b, ws, nblk, Co, Ci), (tw, tb) in zip(jobs, targets):
        base = ws.data_ptr()
        dw = tw is not None
        if tw is None:
            tw = flat[o:o + Co * Ci].view(Co, Ci)
            o += Co * Ci
        table[n] = _lib.ReduceJob(base, tw.data_ptr(), nblk, Co * Ci)
        installs.append((W, tw, dw))
        n += 1
        if b is not None:
            db = tb is not None
            if tb is None:
                tb = flat[o:o + Co]
                o += Co
            table[n] = _lib.ReduceJob(base + 4 * nblk * Co * Ci, tb.data_ptr(), nblk, Co)
            installs.append((b, tb, db))
            n += 1
    # every sum of the pass -- the float weight-gradient slabs and the PointConv layers' float64 slabs -- in ONE launch
    _lib.call('crfconv_reduce_jobs_both', ctypes.cast(table, ctypes.c_void_p), n, None if not n64 else ctypes.cast(arr64, ctypes.c_void_p), n64, st)
    del keep64
    _flush_fold1_bwd()                                  # reads the float64 sums
    _flush_mlp_dw()
    for fn, _ in late_calls:                            # launches that read what the sums above produced
        fn()
    for prm, gr, was_direct in installs:
        _install_grad(prm, gr, was_direct)
    _DEFER['claimed'] = set()


# ------------------------------------------------------------------------------ per-point Linear
_MFMA_MIN_ROWS = 12288      # below this the tiled product (gemm.hip) and the small-MLP nodes; swept on the step: 4096 -> 4.733 ms, 12288 (the 10 240-row level joins the small forms) -> 4.694 ms, 65536 -> 4.736 ms




def _mfma_ok(m, ci, co):
    return m >= _MFMA_MIN_ROWS and bool(_lib.load().crfconv_linear_forward_supported(ci, co))


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


_NO_JOIN_ENV = False      # tests: True = lin_out, bn_apply and add_lrelu as separate nodes (the fused nodes must give the same results)


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
    if not (mlp_block_ok(x, W, None, bn, True) and not _mlp_small_ok(m, ci, co)):
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
    if _NO_JOIN_ENV or table.padded:
        return None
    m, ci = x.shape
    co = W.shape[0]
    if m != table.m_src or not (mlp_block_ok(x, W, None, bn, True) and not _mlp_small_ok(m, ci, co)):
        return None
    require_gpu(x, W)
    tick(bn)
    mom = 0.1 if bn.momentum is None else bn.momentum
    if fork and x.requires_grad and torch.is_grad_enabled() and not _NO_FORK_ENV:
        return _MLPBlockPool.apply(x, W, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps, table, True)
    out = _MLPBlockPool.apply(x, W, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps, table, False)
    return (out, x) if fork else out




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
        _lib.call('crfconv_mlp_small_backward', ptr(gA), ptr(y), ptr(coef), ptr(W), ptr(None if addend is None else addend.contiguous()),
                  m, ci, co, 1, float(slope), ptr(gY), ptr(dX), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, ptr(_ticket(dev)), stream_ptr())
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
    if not _small_mlp_disabled and lib.crfconv_mlp_small_supported(m, ci, co) == 1:
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
    if _NO_JOIN_ENV or skip.shape[:-1] != x.shape[:-1] or skip.shape[-1] != W.shape[0] or skip.dtype != torch.float32:
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


_sync_ws = {}


def gridsync_ws(dev):
    """The barrier words of the one-launch kernels (csrc/gridsync.hpp): zeroed once, left zero by every launch.  One buffer
    per (device, stream): launches that share one are ordered by their stream (see _stream_key for captured graphs)."""
    return _stream_buf(_sync_ws, dev, lambda: torch.zeros(_lib.load().crfconv_gridsync_workspace() // 4, dtype=torch.int32, device=dev))


_small_mlp_disabled = __import__('os').environ.get('CRFCONV_NO_ONE_LAUNCH_MLP') is not None       # set by check_gridsync after a barrier failure (A/B: from the start): launch-separated forward from then on


def _mlp_small_ok(m, ci, co):
    if m >= _MFMA_MIN_ROWS:
        return False
    lib = _lib.load()
    if not _small_mlp_disabled and lib.crfconv_mlp_small_supported(m, ci, co) == 1:
        return True                                          # forward in one launch
    # past the one-launch kernel's co-residency limit (or after a barrier failure): the same autograd nodes, forward as product +
    # BatchNorm launches (_small_fwd), backward as always (_small_bwd)
    return ci % 4 == 0 and co % 4 == 0 and lib.crfconv_mlp_small_backward_supported(m, ci, co) == 1


def fail_word_ptrs(dev):
    """Device addresses of the sticky barrier-failure words of every barrier workspace of `dev` (at most 8: the kernel-side guard's
    capacity; the capture buffer and the most recent streams' first)."""
    want = torch.device(dev).index
    if want is None:
        want = torch.cuda.current_device()
    word = _lib.load().crfconv_gridsync_fail_word()
    keys = [k for k in _sync_ws if k[0] == want]
    keys.sort(key=lambda k: 0 if k[1] == 'capture' else 1)
    return [_sync_ws[k].data_ptr() + 4 * word for k in keys[:8]]


def check_gridsync(dev=None, reduced_flag=None):
    """Raises CrfConvError when a one-launch kernel's grid barrier has timed out on `dev` since the last check (its
    workgroups were not all resident -- CU mask, reserved CUs; the launch's outputs were NaN-poisoned).  One 4-byte
    device read per barrier workspace (a synchronisation): FlatSGD.step() calls it every `check_every` eager steps, a loop that
    replays captured graphs should call it once per epoch / logging interval.  `reduced_flag`: the guard slot of the gradient bucket
    (distributed.FlatGradAllReduce.guard) -- under data parallelism it holds the SUM of the ranks' flags after the all-reduce, so
    every rank raises in the same step, not only the one whose kernel failed (the others would hang in the next collective).
    What survives a failure: PARAMETERS and the MOMENTUM buffer on every rank -- FlatSGD's update kernel reads the same sticky
    words and the reduced slot and changes nothing while one is set (eager steps and captured replays alike), so every step
    since the failure was a no-op for them.  What does not: the BatchNorm RUNNING statistics of the layers downstream of the failed
    launch saw NaN activations in those steps (the failed layer itself skips its update) -- restore the model's buffers from the
    last checkpoint, or reset them, before going on.  After a failure the one-launch KERNEL is switched off for the rest of the
    process (the small-MLP nodes go on with a launch-separated forward: tiled product + BatchNorm launches, _small_fwd), so a
    caller that catches the error and has repaired the buffers can re-run the step."""
    global _small_mlp_disabled
    word = _lib.load().crfconv_gridsync_fail_word()
    bad = []
    want = None if dev is None else torch.device(dev).index
    for table in (_sync_ws,):
        for key, ws in list(table.items()):
            if want is not None and key[0] is not None and key[0] != want:
                continue
            code = int(ws[word].item())
            if code != 0:
                ws.zero_()                                    # barrier counts and the flag: a clean slate for the retry
                bad.append((key, code))
    remote = False
    if reduced_flag is not None and reduced_flag.numel():
        v = float(reduced_flag.reshape(-1)[0].item())
        remote = not (v == 0.0)
        if remote:
            reduced_flag.zero_()
    if bad or remote:
        _small_mlp_disabled = True
        where = ('device(s) %s (code 0x%x)' % (sorted({k[0] for k, _ in bad}), bad[0][1])) if bad else 'another rank of the process group'
        raise _lib.CrfConvError('grid barrier timed out on %s: a one-launch kernel could not get all its '
                                'workgroups resident; its outputs were poisoned with NaN.  The one-launch MLP path is now '
                                'disabled for this process (CRFCONV_NO_ONE_LAUNCH_MLP=1 does the same up front).' % where)


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
        _lib.call('crfconv_mlp_small_backward_jobs', ctypes.cast(jobs, ctypes.c_void_p), n, ptr(_ticket(per[0][0].device)), st)
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
    if shared and (_NO_FORK_ENV or blocks[0][0] is not blocks[1][0]):
        return None                                    # (tests: the un-forked graph runs the blocks one by one)
    for i, (x, W, bn, slope, fork) in enumerate(blocks):
        tick(bn)
        mom = 0.1 if bn.momentum is None else bn.momentum
        use_fork = bool(fork and x.requires_grad and torch.is_grad_enabled() and not _NO_FORK_ENV)
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
    if not (ca % 4 == 0 and cb % 4 == 0 and xb.dtype == torch.float32 and m >= _MFMA_MIN_ROWS
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


_NO_FORK_ENV = False      # tests: True = autograd's own accumulation pass instead of the fork chain


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
    if fork and x2.requires_grad and torch.is_grad_enabled() and not _NO_FORK_ENV:
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


# ------------------------------------------------------------------------------ PointConv
class MomentsEntry(tuple):
    """relpos_moments(...) as a tuple plus the inputs it came from: ``refresh_`` recomputes it INTO the same tensors
    (NeighborTable.refresh_ and the version check of PointConv._moments call it), so kernels launched from a captured
    graph read the statistics of the batch that is in the buffers now."""

    def __new__(cls, values, pos_src, pos_tgt):
        obj = super().__new__(cls, values)
        obj.pos_src, obj.pos_tgt = pos_src, pos_tgt
        obj.versions = (pos_src._version, pos_tgt._version)
        return obj

    def refresh_(self, table):
        relpos_moments(self.pos_src, self.pos_tgt, table, out=self)       # in place: no temporaries, no copies
        self.versions = (self.pos_src._version, self.pos_tgt._version)

    def batch_job(self, table):
        """This entry's refresh as one job of crfconv_pointconv_moments_batched (graph.batched_reverse issues them together)."""
        return _lib.MomentsJob(self.pos_src.data_ptr(), self.pos_tgt.data_ptr(), table.idx32.data_ptr(), table.K, table.m_tgt,
                               float(table.n_edges), self[0].data_ptr(), self[1].data_ptr(), self[3].data_ptr(), self[4].data_ptr())

    def mark_fresh(self):
        self.versions = (self.pos_src._version, self.pos_tgt._version)

    def stale(self):
        return self.versions != (self.pos_src._version, self.pos_tgt._version)


def relpos_moments(pos_src, pos_tgt, table, out=None):
    """(mean [3], covariance [3,3], edge count, packed float64 [12], mean float32 [3]) of rel = p_tgt[i] - p_src[j]
    over all edges: one pass over the edges, one finishing launch (crfconv_pointconv_moments_packed).  ``out``: an earlier
    result whose tensors are overwritten in place (refresh of a static batch)."""
    require_gpu(pos_src, pos_tgt)
    dev = pos_src.device
    n = float(table.n_edges)
    if out is not None:
        mean, cov, packed, mean32 = out[0], out[1], out[3], out[4]
    else:
        mean = torch.empty(3, dtype=torch.float64, device=dev)
        cov = torch.empty((3, 3), dtype=torch.float64, device=dev)
        packed = torch.empty(12, dtype=torch.float64, device=dev)
        mean32 = torch.empty(3, dtype=torch.float32, device=dev)
    nbytes = _lib.load().crfconv_pointconv_workspace(table.m_tgt, table.K, 4)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.call('crfconv_pointconv_moments_packed', ptr(pos_src), ptr(pos_tgt), ptr(table.idx32), table.K, table.m_tgt, n,
              ptr(mean), ptr(cov), ptr(packed), ptr(mean32), ptr(ws), nbytes, stream_ptr())
    return mean, cov, n, packed, mean32                   # [4]: mean rel in float32 (kernel argument)


class _PointConv(torch.autograd.Function):
    """out[i,c] = sum_k w_ik[c] x[j,c],  w = BN2(W2 lrelu(BN1(W1 rel))),  rel = p_tgt[i] - p_src[j].

    Both BatchNorms are folded into per-channel coefficients by tiny kernels (fold1 / fold2: BN-1's batch
    statistics are analytic in the moments of rel, BN-2's come from one reduction pass over the edges);
    the backward mirrors it: reduction pass -> fold2_bwd -> parameter pass -> fold1_bwd, plus the
    source-major gather for dx.  Nothing per-edge is ever stored (except for d >= 64, see bwd_dump)."""

    @staticmethod
    def forward(ctx, x, W1, g1, be1, W2, g2, be2, pos_src, pos_tgt, table, mom, bn1_state, bn2_state, slope, mom32=None, prefold=None):
        require_gpu(x, W1, W2, pos_src, pos_tgt)
        if x.shape[0] != table.m_src or pos_src.shape[0] != table.m_src or pos_tgt.shape[0] != table.m_tgt:
            raise _lib.CrfConvError('point_conv: x %d / pos_src %d rows for %d sources, pos_tgt %d rows for %d targets'
                                    % (x.shape[0], pos_src.shape[0], table.m_src, pos_tgt.shape[0], table.m_tgt))
        dev = x.device
        x, W1c, W2c = _f32c(x), _f32c(W1), _f32c(W2)
        g1c, be1c, g2c, be2c = _f32c(g1), _f32c(be1), _f32c(g2), _f32c(be2)
        d = x.shape[1]
        m_tgt, K = table.m_tgt, table.K
        st = stream_ptr()
        lib = _lib.load()
        nbytes = lib.crfconv_pointconv_workspace(max(m_tgt, table.m_src), K, d)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        n_e = float(table.n_edges)
        use1, rm1, rv1, mom1, eps1 = bn1_state
        use2, rm2, rv2, mom2, eps2 = bn2_state
        if prefold is not None:                            # BatchNorm-1 folded for all layers of the network in one launch
            A1, b1, aux1 = prefold                         # (point_conv_prefold), incl. the running-statistics update
        else:
            A1 = torch.empty((d, 3), dtype=torch.float32, device=dev)
            b1 = torch.empty(d, dtype=torch.float32, device=dev)
            aux1 = torch.empty(3 * d, dtype=torch.float64, device=dev)
            _lib.call('crfconv_pointconv_fold1', ptr(W1c), ptr(g1c), ptr(be1c), ptr(mom), n_e, ptr(rm1), ptr(rv1),
                      float(mom1), float(eps1), 1 if use1 else 0, d, ptr(A1), ptr(b1), ptr(aux1), st)
        shift = (torch.empty if use2 else torch.zeros)(d, dtype=torch.float32, device=dev)
        stats = U = V = None
        if use2:
            # batch statistics of h2, U = sum_k (h2 - shift) x_j and V = sum_k x_j from ONE pass over the edges
            stats = torch.empty(2 * d, dtype=torch.float64, device=dev)
            U = torch.empty((m_tgt, d), dtype=torch.float32, device=dev)
            V = torch.empty((m_tgt, d), dtype=torch.float32, device=dev)
            mean_rel = mom32 if mom32 is not None else mom[:3].float()
            riders = _take_riders(K, d)
            if riders is not None:                   # the CRF layers' matrices ride along in this launch (crf_matrices_batched(ride=True))
                rc, rH, rQ, rP = riders
                _lib.call('crfconv_pointconv_forward_uv_hosting', ptr(x), ptr(pos_src), ptr(pos_tgt), ptr(table.idx32), K, m_tgt, d,
                          ptr(A1), ptr(b1), ptr(W2c), slope, ptr(mean_rel), ptr(shift), ptr(stats), ptr(U), ptr(V), ptr(ws),
                          nbytes, _pc_ticket(dev), _ptr_array(rc), rH, len(rc), _ptr_array(rQ), _ptr_array(rP), st)
            else:
                _lib.call('crfconv_pointconv_forward_uv', ptr(x), ptr(pos_src), ptr(pos_tgt), ptr(table.idx32), K, m_tgt, d,
                          ptr(A1), ptr(b1), ptr(W2c), slope, ptr(mean_rel), ptr(shift), ptr(stats), ptr(U), ptr(V), ptr(ws),
                          nbytes, _pc_ticket(dev), st)
        a2 = torch.empty(d, dtype=torch.float32, device=dev)
        b2 = torch.empty(d, dtype=torch.float32, device=dev)
        aux2 = torch.empty(2 * d, dtype=torch.float64, device=dev)
        out = torch.empty((m_tgt, d), dtype=torch.float32, device=dev)
        if use2:       # BatchNorm-2 folded from the statistics inside the elementwise combine
            _lib.call('crfconv_pointconv_combine', ptr(U), ptr(V), ptr(stats), ptr(shift), ptr(g2c), ptr(be2c), n_e,
                      ptr(rm2), ptr(rv2), float(mom2), float(eps2), m_tgt, d, ptr(a2), ptr(b2), ptr(aux2), ptr(out), st)
        else:
            _lib.call('crfconv_pointconv_fold2', ptr(stats), ptr(shift), ptr(g2c), ptr(be2c), n_e, ptr(rm2), ptr(rv2),
                      float(mom2), float(eps2), 0, d, ptr(a2), ptr(b2), ptr(aux2), st)
            _lib.call('crfconv_pointconv_forward', ptr(x), ptr(pos_src), ptr(pos_tgt), ptr(table.idx32), K, m_tgt, d,
                      ptr(A1), ptr(b1), ptr(W2c), slope, ptr(a2), ptr(b2), ptr(out), st)
        ctx.uv = (U, V)
        ctx.prm = (W1, g1, be1, W2)                        # the parameter objects themselves (deferred / direct gradients)
        ctx.table, ctx.n_e, ctx.slope, ctx.use1, ctx.use2, ctx.eps1 = table, n_e, slope, use1, use2, eps1
        ctx.save_for_backward(x, W1c, g1c, W2c, g2c, A1, b1, a2, b2, shift, aux1, aux2, mom, pos_src, pos_tgt)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, W1, g1, W2, g2, A1, b1, a2, b2, shift, aux1, aux2, mom, pos_src, pos_tgt = ctx.saved_tensors
        table, n_e, slope = ctx.table, ctx.n_e, ctx.slope
        pW1, pg1, pbe1, pW2 = ctx.prm
        dev = x.device
        d = x.shape[1]
        m_tgt, K = table.m_tgt, table.K
        g = _f32c(gout)
        st = stream_ptr()
        nbytes = _lib.load().crfconv_pointconv_workspace(max(m_tgt, table.m_src), K, d)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        # pass 1: sum g_w and sum g_w (h2 - shift)  ->  BatchNorm-2 backward coefficients
        coef = torch.empty((5, d), dtype=torch.float32, device=dev)       # ca, cb, cc, dgamma2, dbeta2
        U, V = ctx.uv
        rev_ptr, rev_eid = table.reverse
        dx = torch.empty((table.m_src, d), dtype=torch.float32, device=dev)
        if U is not None:                  # training forward left U, V: the reductions are row sums, no edge pass --
            # and they ride in the input-gradient launch (source-major gather over the reverse table), which needs none of their results
            _lib.call('crfconv_pointconv_bwd_input_reduce', ptr(g), ptr(pos_src), ptr(pos_tgt), ptr(rev_ptr), ptr(rev_eid), K, table.m_src, m_tgt, d,
                      ptr(A1), ptr(b1), ptr(W2), slope, ptr(a2), ptr(b2), ptr(dx), ptr(U), ptr(V), ptr(shift), ptr(aux2), ptr(g2), n_e,
                      1 if ctx.use2 else 0, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), ptr(coef[4]), ptr(ws), nbytes,
                      _pc_ticket(dev), st)       # (its partial rows in `ws` are consumed inside the launch; the parameter pass reuses `ws`)
        else:
            _lib.call('crfconv_pointconv_bwd_input', ptr(g), ptr(pos_src), ptr(pos_tgt), ptr(rev_ptr), ptr(rev_eid), K,
                      table.m_src, d, ptr(A1), ptr(b1), ptr(W2), slope, ptr(a2), ptr(b2), ptr(dx), st)
            red = torch.empty(2 * d, dtype=torch.float64, device=dev)
            _lib.call('crfconv_pointconv_bwd_reduce', ptr(x), ptr(g), ptr(pos_src), ptr(pos_tgt), ptr(table.idx32), K,
                      m_tgt, d, ptr(A1), ptr(b1), ptr(W2), slope, ptr(shift), ptr(red), ptr(ws), nbytes, st)
            _lib.call('crfconv_pointconv_fold2_bwd', ptr(red), ptr(shift), ptr(aux2), ptr(g2), n_e, 1 if ctx.use2 else 0,
                      d, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), ptr(coef[4]), st)
        # pass 2: parameter gradients
        dW2_64 = None
        # the fold of this layer's parameter gradients waits for the end of the backward pass (below): then so can the SUMS of their
        # partial slabs -- one crfconv_reduce_jobs_f64 launch for all PointConv layers instead of two / one per layer
        late = all(_defer_ok((q, None)) for q in (pW1, pg1, pbe1, pW2))
        if d <= _PC_PARAMS_INKERNEL_MAX_D:
            dW2 = torch.empty(d * d, dtype=torch.float64, device=dev)
            dA1b1 = torch.empty((d, 4), dtype=torch.float64, device=dev)
            _lib.call('crfconv_pointconv_bwd_params', ptr(x), ptr(g), ptr(pos_src), ptr(pos_tgt), ptr(table.idx32),
                      K, m_tgt, d, ptr(A1), ptr(b1), ptr(W2), slope, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]),
                      None if late else ptr(dW2), None if late else ptr(dA1b1), ptr(ws), nbytes, st)
            if late:
                sw, sa, nb = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
                _lib.call('crfconv_pointconv_bwd_params_slabs', ptr(ws), m_tgt, d, ctypes.byref(sw), ctypes.byref(sa), ctypes.byref(nb))
                _defer_reduce64(sw.value, True, nb.value, d * d, dW2, (ws,))
                _defer_reduce64(sa.value, False, nb.value, 4 * d, dA1b1, (ws,))
            dW2_64, dW2 = dW2, torch.empty((d, d), dtype=torch.float32, device=dev)      # cast by the fold kernel below
        elif late:
            # wide, edge-poor levels: the WHOLE parameter pass (per-edge dump, g_h2^T h1 partials, g_h1 = g_h2 W2, dA1 | db1 slabs) waits
            # for the end of the backward pass, where the passes of all wide layers go out as a handful of launches (_flush_pc_wide)
            dA1b1 = torch.empty((d, 4), dtype=torch.float64, device=dev)
            _DEFER['pc_wide'].append(dict(x=x, g=g, pos_src=pos_src, pos_tgt=pos_tgt, idx=table.idx32, K=K, m_tgt=m_tgt, d=d, A1=A1, b1=b1,
                                          W2=W2, slope=slope, coef=coef, pW2=pW2, dA1b1=dA1b1))
            _arm_flush()
            dW2 = None
        else:
            # wide, edge-poor levels: per-edge h1 / g_h2 / rel to HBM, contractions as dense GEMMs
            E = m_tgt * K
            h1 = torch.empty((E, d), dtype=torch.float32, device=dev)
            gh2 = torch.empty((E, d), dtype=torch.float32, device=dev)
            rel = torch.empty((E, 3), dtype=torch.float32, device=dev)
            _lib.call('crfconv_pointconv_bwd_dump', ptr(x), ptr(g), ptr(pos_src), ptr(pos_tgt), ptr(table.idx32), K,
                      m_tgt, d, ptr(A1), ptr(b1), ptr(W2), slope, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(h1),
                      ptr(gh2), ptr(rel), st)
            if _defer_ok((pW2, None)):                                     # g_h2^T h1: partials now, the reduction with all others
                _defer_weight_grad(gh2, h1, (pW2, None), False)
                dW2 = None
            else:
                dW2 = torch.empty((d, d), dtype=torch.float32, device=dev)   # ... on the MFMA row-reduction kernel
                wbytes = _lib.load().crfconv_linear_wgrad_workspace(E, d, d)
                wws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
                _lib.call('crfconv_linear_wgrad', ptr(gh2), ptr(h1), E, d, d, ptr(dW2), None, ptr(wws), wbytes, st)
            gw = _gemm(gh2, W2)                                            # g_h1 before the LeakyReLU mask
            dA1b1 = torch.empty((d, 4), dtype=torch.float64, device=dev)
            abytes = _lib.load().crfconv_pointconv_bwd_a1_workspace(E, d)
            aws = torch.empty(abytes, dtype=torch.uint8, device=dev)
            _lib.call('crfconv_pointconv_bwd_a1', ptr(gw), ptr(h1), ptr(rel), E, d, slope, None if late else ptr(dA1b1), ptr(aws),
                      abytes, st)
            if late:
                slab = (aws.data_ptr() + 255) & ~255
                _defer_reduce64(slab, False, _lib.load().crfconv_pointconv_bwd_a1_nblk(E, d), 4 * d, dA1b1, (aws,))
        defer_fold = all(_defer_ok((q, None)) for q in (pW1, pg1, pbe1)) and (dW2_64 is None or _defer_ok((pW2, None)))
        if defer_fold:
            # nothing in this pass reads dW1 / dgamma1 / dbeta1 (or the float32 dW2 of the narrow layers): ONE batched fold launch
            # for all PointConv layers at the end of the backward, written straight into the caller's bucket where there is one
            outs = [(q,) + _param_out(q, shp, dev) for q, shp in ((pW1, (d, 3)), (pg1, (d,)), (pbe1, (d,)))]
            if dW2_64 is not None:
                outs.append((pW2,) + _param_out(pW2, (d, d), dev))
            adr = lambda t: None if t is None else t.data_ptr()
            job = _lib.Fold1BwdJob(adr(W1), adr(g1), adr(mom), adr(aux1), adr(dA1b1), float(ctx.eps1), 1 if ctx.use1 else 0, d, 0,
                                   adr(outs[0][1]), adr(outs[1][1]), adr(outs[2][1]), adr(dW2_64),
                                   adr(outs[3][1]) if dW2_64 is not None else None)
            _defer_fold1_bwd(job, (W1, g1, mom, aux1, dA1b1, dW2_64), outs)
            dW1 = dg1 = dbe1 = None
            if dW2_64 is not None:
                dW2 = None
        else:
            dW1 = torch.empty((d, 3), dtype=torch.float32, device=dev)
            dg1 = torch.empty(d, dtype=torch.float32, device=dev)
            dbe1 = torch.empty(d, dtype=torch.float32, device=dev)
            _lib.call('crfconv_pointconv_fold1_bwd', ptr(W1), ptr(g1), ptr(mom), ptr(aux1), ptr(dA1b1), float(ctx.eps1),
                      1 if ctx.use1 else 0, d, ptr(dW1), ptr(dg1), ptr(dbe1), ptr(dW2_64), ptr(dW2) if dW2_64 is not None else None, st)
        return (dx, dW1, dg1, dbe1, dW2, coef[3], coef[4], None, None, None, None, None, None, None, None, None)


_PC_D = (4, 8, 16, 32, 64, 128)
_PC_PARAMS_INKERNEL_MAX_D = 16          # wider: the matrix-pipe parameter pass (d = 32, 64) / per-edge dump + MFMA reductions (d = 128)


def pack_moments(moments):
    """(mean [3], cov [3,3], n) -> 12 float64 {mean, cov row-major} for the fold kernels."""
    mean, cov = moments[0], moments[1]
    return torch.cat([mean.reshape(3), cov.reshape(9)]).contiguous()


_NO_PREFOLD_ENV = False      # tests: True = one fold launch inside every PointConv layer


def _bn_state(bn, training, momentum, advance=True):
    """(use batch statistics, running mean / var to read or update -- or None --, momentum, eps) of one BatchNorm1d."""
    use_batch = training or bn.running_mean is None
    if training and advance:
        tick(bn)
    upd = training and bn.running_mean is not None
    keep = upd or not use_batch
    return (use_batch, bn.running_mean if keep else None, bn.running_var if keep else None,
            momentum if bn.momentum is None else bn.momentum, bn.eps)


def point_conv_prefold(layers, training, momentum=0.1):
    """BatchNorm-1 folding (crfconv_pointconv_fold1) of SEVERAL PointConv layers in one launch.  Everything it reads -- the
    first weight-MLP layer and the rel-pos moments of the layer's table -- exists before the forward pass starts, so a
    network folds all its layers up front (models/point_conv_big.py:113-131 has ten) instead of one tiny launch inside every
    layer.  layers: (W1 [d, 3], bn1, moments, table) per layer; returns one ``prefold`` for each, to be passed to point_conv
    (which then skips its own fold; it still advances the BatchNorm's step counter).  Running statistics are updated here."""
    if not layers:
        return []
    dev = layers[0][0].device
    ds = [int(W1.shape[0]) for W1, _, _, _ in layers]
    tot = sum(ds)
    A1 = torch.empty((tot, 3), dtype=torch.float32, device=dev)
    b1 = torch.empty(tot, dtype=torch.float32, device=dev)
    aux1 = torch.empty(3 * tot, dtype=torch.float64, device=dev)
    jobs, keep, out, o = [], [], [], 0
    adr = lambda t: None if t is None else t.data_ptr()
    for (W1, bn1, moments, table), d in zip(layers, ds):
        require_gpu(W1)
        mom = moments[3] if len(moments) > 3 else pack_moments(moments)
        use1, rm1, rv1, mom1, eps1 = _bn_state(bn1, training, momentum, advance=False)
        W1c, g1c, be1c = _f32c(W1), _f32c(bn1.weight), _f32c(bn1.bias)
        pre = (A1[o:o + d], b1[o:o + d], aux1[3 * o:3 * o + 3 * d])
        jobs.append(_lib.Fold1Job(adr(W1c), adr(g1c), adr(be1c), adr(mom), float(table.n_edges), adr(rm1), adr(rv1), float(mom1),
                                  float(eps1), 1 if use1 else 0, d, adr(pre[0]), adr(pre[1]), adr(pre[2])))
        keep.append((W1c, g1c, be1c, mom))
        out.append(pre)
        o += d
    table_ = (_lib.Fold1Job * len(jobs))(*jobs)
    _lib.call('crfconv_pointconv_fold1_batched', ctypes.cast(table_, ctypes.c_void_p), len(jobs), stream_ptr())
    return out


def point_conv(x, pos_src, pos_tgt, table, W1, bn1, W2, bn2, training, momentum=0.1, moments=None, slope=0.1, prefold=None):
    """Functional PointConv over flattened clouds.

    x [m_src, d]; pos_* [m, 3]; W1 [d, 3], W2 [d, d] Linear weights (no bias);
    bn1 / bn2: torch.nn.BatchNorm1d modules (affine + running statistics, updated in training);
    prefold: this layer's entry of point_conv_prefold (same W1, bn1, moments, table and mode), or None."""
    d = x.shape[1]
    if d not in _PC_D:
        raise _lib.CrfConvError('PointConv width d=%d not in %s' % (d, _PC_D))
    pos_src = _f32c(pos_src)
    pos_tgt = pos_src if pos_tgt is None else _f32c(pos_tgt)
    if moments is None:
        moments = relpos_moments(pos_src, pos_tgt, table)
    mom = moments[3] if len(moments) > 3 else pack_moments(moments)
    mom32 = moments[4] if len(moments) > 4 else None
    return _PointConv.apply(x, W1, bn1.weight, bn1.bias, W2, bn2.weight, bn2.bias, pos_src, pos_tgt, table, mom,
                            _bn_state(bn1, training, momentum), _bn_state(bn2, training, momentum), float(slope), mom32, prefold)


__all__ = ['linear', 'bn_act', 'crf_meanfield', 'gather_rows', 'neighbor_maxpool', 'relpos_moments', 'point_conv',
           'point_conv_prefold', 'cross_entropy', 'training_loss', 'NeighborTable']
