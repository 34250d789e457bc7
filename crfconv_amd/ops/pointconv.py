"""PointConv: rel-pos moments, the layer node, the batched BatchNorm-1 prefold."""
import ctypes

import torch

from .. import _lib
from ..graph import NeighborTable, ptr, require_gpu, stream_ptr
from ._base import _f32c, _pc_ticket, _ptr_array

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


# names of the sibling modules, imported LAST: every use is inside a function body, so import cycles between the families are harmless
from .defer import _DEFER, _arm_flush, _defer_fold1_bwd, _defer_ok, _defer_reduce64, _defer_weight_grad, _param_out  # noqa: E402
from .dense import _gemm, tick  # noqa: E402
from .crf import _take_riders  # noqa: E402
