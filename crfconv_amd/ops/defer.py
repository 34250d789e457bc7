"""Deferred parameter work of a backward pass: queues and the batched launches at its end."""
import ctypes

import torch

from .. import _lib
from ..graph import NeighborTable, ptr, require_gpu, stream_ptr
from ._base import state

# ------------------------------------------------------------------------------ deferred weight gradients
# Every Linear's dW = G^T X ends in a small "sum the row-slice partials" launch; PointConvBig has 74 of them per
# backward pass, each far below the cost of launching it.  Inside ``with deferred_weight_grads():`` the MFMA kernel
# only writes its partials, and ONE batched launch at the end of the backward pass (an autograd engine callback,
# like DDP's) finishes all of them and installs / accumulates ``.grad`` of the weight and bias parameters directly.
# Opt-in because it bypasses autograd for those leaves: ``torch.autograd.grad(loss, weight)`` sees nothing, and
# gradient hooks on the weights do not fire.  ``loss.backward()`` + ``param.grad`` behave as usual.
_DEFER = {'on': False, 'jobs': [], 'partials': [], 'red64': [], 'pc_wide': [], 'tn': [], 'late_calls': [], 'late_mats': [], 'folds': [], 'mlpdw': [], 'armed': False,
          'claimed': set()}


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
            for k in ('partials', 'red64', 'pc_wide', 'tn', 'late_calls', 'late_mats'):
                _DEFER[k] = []
        return self

    def __exit__(self, exc_type, *exc):
        _DEFER['on'] = self.prev
        _DEFER['sink'] = self.prev_sink
        if exc_type is not None and not self.prev:
            # the backward raised after arming the engine callback: drop its queued partials, or every later backward
            # would find 'armed' set, never queue the callback again and silently lose all Linear weight gradients
            _DEFER['jobs'], _DEFER['folds'], _DEFER['mlpdw'], _DEFER['armed'], _DEFER['claimed'] = [], [], [], False, set()
            for k in ('partials', 'red64', 'pc_wide', 'tn', 'late_calls', 'late_mats'):
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
    """The weight gradients of all fused MLP blocks (crfconv_mlp_dw_jobs) and the queued backward of the CRF layers' matrices
    (ops.crf: 'late_mats' -- it reads the dP / dQ sums of the launch in front) -- as ONE launch when both are there: the matrices'
    slabs ride as the first workgroups of the weight-gradient launch (crfconv_mlp_dw_jobs_hosting)."""
    jobs, _DEFER['mlpdw'] = _DEFER['mlpdw'], []
    mats, _DEFER['late_mats'] = _DEFER.get('late_mats', []), []
    st = stream_ptr()
    if jobs:
        table = (_lib.MlpDwJob * len(jobs))(*[j[0] for j in jobs])
        if mats and state.dw_hosts_mats:
            arguments, install, _ = mats.pop(0)
            arrays, keep = arguments()
            _lib.call('crfconv_mlp_dw_jobs_hosting', ctypes.cast(table, ctypes.c_void_p), len(jobs), *arrays, st)
            del keep
            install()
        else:
            _lib.call('crfconv_mlp_dw_jobs', ctypes.cast(table, ctypes.c_void_p), len(jobs), st)
        for _, _, (prm, gr, direct) in jobs:
            _install_grad(prm, gr, direct)
    for arguments, install, _ in mats:
        arrays, keep = arguments()
        _lib.call('crfconv_crf_matrices_backward_batched', *arrays, st)
        del keep
        install()


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
