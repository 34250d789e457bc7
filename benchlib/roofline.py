"""Roofline objects of the bench line: the level-0 mean-field forward / backward, the layer as the network runs it, PointConv, and the
algorithmic byte model of the whole step."""
import numpy as np
import torch

from .common import (HBM_PEAK, ROCPROF_MF, TRAFFIC_BWD, TRAFFIC_FWD, TRAFFIC_PC, TRAFFIC_STEP, _event_time, _meanfield_problem, _measured_traffic,
                     _rocprof_durations)


def roofline_meanfield(data, dev, H=8, T=3, level=0, form=None):
    """CRF mean-field forward of one level alone (level 0 = the kernel the north_star target is stated on), HIP-event timed on
    the stream it is launched on.  form: 'block' (one launch, block-resident rows: csrc/crf_block.hip) / 'steps' (one launch per
    step: csrc/crf.hip) / None = what ops.crf_meanfield picks for this table (block where the shape is covered and the table is
    local); the other form's time is reported beside it."""
    from crfconv_amd import _lib
    from crfconv_amd.graph import ptr, stream_ptr
    from crfconv_amd.ops._base import gridsync_ws
    from crfconv_amd.ops.crf import _block_rows
    tab, m, K, z, y, Q, P, _ = _meanfield_problem(data, dev, H, level=level)
    s = torch.empty(m, K, device=dev)
    xs = torch.empty(T, m, H, device=dev)
    st = stream_ptr()
    ws = gridsync_ws(dev)
    can_block = _lib.load().crfconv_meanfield_forward_block_rows(m, H, K, 1, T) > 0
    if form is None:
        _block_rows(tab, m, H, 1, T)            # (the first question about a table only starts the locality measurement: ops.crf._table_is_local)
        form = 'block' if _block_rows(tab, m, H, 1, T) > 0 else 'steps'
    elif form == 'block' and not can_block:
        return None

    def launch_steps():
        _lib.call('crfconv_meanfield_forward_u16', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src,
                  K, 1, m, H, ptr(Q), ptr(P), T, ptr(s), ptr(xs), st)

    def launch_block():
        _lib.call('crfconv_meanfield_forward_block', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src,
                  K, 1, m, H, ptr(Q), ptr(P), T, ptr(s), ptr(xs), ptr(ws), st)
    launch = launch_block if form == 'block' else launch_steps
    avg, lo = _event_time(launch)
    other = None
    if level == 0 and (form == 'steps' and can_block or form == 'block'):
        other = _event_time(launch_steps if form == 'block' else launch_block)[0]
    alg_bytes = m * (4 * (K - 1) + 4 * H * (2 * T + 1))
    traffic, note = _measured_traffic(TRAFFIC_FWD, {'m': m, 'H': H, 'K': K, 'T': T, 'u16': tab.idx16 is not None})
    out = {'bound': 'hbm', 'achieved': alg_bytes / avg / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
           'frac': alg_bytes / avg / HBM_PEAK, 'traffic': traffic, 'traffic_source': note,
           'kernel': ('crfconv_meanfield_forward_block level-%d (mf_block_kernel: ONE launch, block-resident rows, %d grid barriers, m=%d, H=%d, K=%d)'
                      % (level, T - 1, m, H, K)) if form == 'block' else
                     ('crfconv_meanfield_forward level-%d (sim_step_fast_kernel [similarity + step 1] + %d x step_fast_kernel, '
                      'm=%d, H=%d, K=%d)' % (level, T - 1, m, H, K)),
           'form': form, 'other_form_launch_us': None if other is None else other * 1e6,
           'block_locality': tab.cache.get(('block_locality', 640)),
           'alg_bytes_per_launch': alg_bytes, 'avg_launch_us': avg * 1e6, 'min_launch_us': lo * 1e6,
           'note': 'isolated synthetic problem, 10 back-to-back launches per event pair: the %.1f MB working set stays resident in '
                   'the 256 MiB Infinity Cache between launches (as it does between the consecutive kernels of the real step, '
                   'whose in-step times agree); peak = the 8 TB/s HBM3E figure' % (alg_bytes / 1e6)}
    if level == 0:
        rec, why = _rocprof_durations({'m': m, 'H': H, 'K': K, 'T': T})
        key = 'fwd_us' if form == 'block' else 'fwd_steps_us'
        out['frac_rocprof'] = None if rec is None else alg_bytes / (rec[key] * 1e-6) / HBM_PEAK
        out['rocprof_launch_us'] = None if rec is None else rec[key]
        out['rocprof_other_form_launch_us'] = None if rec is None else rec['fwd_steps_us' if form == 'block' else 'fwd_us']
        out['frac_rocprof_source'] = why + ' (sum of the average durations of the kernels of this form; frac = HIP events of this run: the profiler adds ~1 us per dispatch)'
    return out


def roofline_meanfield_bwd(data, dev, H=8, T=3, level=0):
    """Level-0 CRF mean-field BACKWARD (crfconv_meanfield_backward, csrc/crf_bwd.hip: T - 1 reverse walks | edge pass over
    all steps + softmax backward | last reverse walk with the dy scatter and the dP / dQ reduction), HIP-event timed.  Algorithmic bytes per point:
    twice the forward's compulsory bytes plus the reverse index, 2 (4 (K-1) + 4 H (2 T + 1)) + 4 K + 4."""
    from crfconv_amd import _lib, ops
    from crfconv_amd.graph import ptr, stream_ptr
    tab, m, K, z, y, Q, P, g = _meanfield_problem(data, dev, H, level=level)
    lib = _lib.load()
    if lib.crfconv_meanfield_backward_supported(H, K, 1) != 1:
        return None
    # H >= 32: dP / dQ are not formed inside the last walk; the launches leave m_t and sum_t G_t for the row-reduction
    # kernel (in the training step those partial passes ride in the batched weight-gradient launches at the end of the pass)
    inside = lib.crfconv_meanfield_backward_param_grads_inside(H) == 1
    mts = None if inside else torch.empty(T, m, H, device=dev)
    sumG = None if inside else torch.empty(m, H, device=dev)
    rev_ptr, rev_eid = tab.reverse
    gout = torch.randn(m, H, generator=g).to(dev)
    s = torch.empty(m, K, device=dev)
    xs = torch.empty(T, m, H, device=dev)
    st = stream_ptr()
    _lib.call('crfconv_meanfield_forward_u16', ptr(z), ptr(y), ptr(tab.idx32), ptr(tab.idx16), tab.n_tgt, tab.n_src,
              K, 1, m, H, ptr(Q), ptr(P), T, ptr(s), ptr(xs), st)
    Gs, dzq = torch.empty(T, m, H, device=dev), torch.empty(m, H, device=dev)
    dz, dy_self, dy = (torch.empty(m, H, device=dev) for _ in range(3))
    w = torch.empty(m, K, device=dev)
    dP, dQ = torch.empty(H, H, device=dev), torch.empty(H, H, device=dev)
    wsb = lib.crfconv_meanfield_backward_workspace(m, H, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ticket = ops._ticket(dev)

    def launch():
        _lib.call('crfconv_meanfield_backward', ptr(gout), ptr(z), ptr(y), ptr(s), ptr(xs), ptr(tab.idx32), ptr(tab.idx16),
                  tab.n_tgt, tab.n_src, ptr(rev_ptr), ptr(rev_eid), K, 1, m, H, ptr(Q), ptr(P), T, ptr(Gs), ptr(dzq), ptr(mts),
                  ptr(sumG), ptr(dz), ptr(w), ptr(dy_self), ptr(dy), ptr(dP), ptr(dQ), ptr(ws), wsb, ptr(ticket), st)
    avg, lo = _event_time(launch, per=5)
    alg_bytes = m * (2 * (4 * (K - 1) + 4 * H * (2 * T + 1)) + 4 * K + 4)
    traffic, note = _measured_traffic(TRAFFIC_BWD, {'m': m, 'H': H, 'K': K, 'T': T})
    out = {'bound': 'hbm', 'achieved': alg_bytes / avg / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
           'frac': alg_bytes / avg / HBM_PEAK, 'traffic': traffic, 'traffic_source': note,
           'kernel': 'crfconv_meanfield_backward level-%d (%d x bwd_rev<chain> + bwd_edge_all + bwd_rev<final>, '
                     'm=%d, H=%d, K=%d)' % (level, T - 1, m, H, K),
           'alg_bytes_per_launch': alg_bytes, 'avg_launch_us': avg * 1e6, 'min_launch_us': lo * 1e6}
    if level == 0:
        rec, why = _rocprof_durations({'m': m, 'H': H, 'K': K, 'T': T})
        out['frac_rocprof'] = None if rec is None else alg_bytes / (rec['bwd_us'] * 1e-6) / HBM_PEAK
        out['rocprof_launch_us'] = None if rec is None else rec['bwd_us']
        out['frac_rocprof_source'] = why
    return out


def roofline_layer(data, dev, T=3, level0=None, backward=True):
    """The mean-field layer AS THE NETWORK RUNS IT: all four decoder levels (deconv1..deconv4: H = 8, 16, 32, 64 on
    m = 163 840 ... 2 560 points at config 2), forward and forward + backward, algorithmic bytes of SURVEY 8(d) summed over the
    levels (81.9 MB forward at config 2) against the summed times.  Each level is timed like `roofline` (isolated problem, HIP
    events, back-to-back launches); levels 1-3 hold 6 % of the points but are launch-latency chains, so the layer figure is far
    below the level-0 one -- that is the point of reporting it."""
    per, tf, tb, af, ab = [], 0.0, 0.0, 0, 0
    for level in range(min(4, len(data.multiscale) - 1)):
        H = 8 << level
        f = level0[0] if (level == 0 and level0) else roofline_meanfield(data, dev, H, T, level=level)
        if not backward:                                  # an inference configuration: the forward layer alone
            per.append({'level': level, 'H': H, 'm': int(np.prod(data.multiscale[level].pos.shape[:2])), 'fwd_us': f['avg_launch_us'],
                        'fwd_alg_bytes': f['alg_bytes_per_launch']})
            tf += f['avg_launch_us'] * 1e-6
            af += f['alg_bytes_per_launch']
            continue
        b = level0[1] if (level == 0 and level0) else roofline_meanfield_bwd(data, dev, H, T, level=level)
        if b is None:
            return None
        per.append({'level': level, 'H': H, 'm': int(np.prod(data.multiscale[level].pos.shape[:2])), 'fwd_us': f['avg_launch_us'],
                    'bwd_us': b['avg_launch_us'], 'fwd_alg_bytes': f['alg_bytes_per_launch'], 'bwd_alg_bytes': b['alg_bytes_per_launch']})
        tf += f['avg_launch_us'] * 1e-6
        tb += b['avg_launch_us'] * 1e-6
        af += f['alg_bytes_per_launch']
        ab += b['alg_bytes_per_launch']
    if not backward:
        return {'bound': 'hbm', 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'fwd_alg_bytes': af, 'fwd_us': tf * 1e6, 'fwd_frac': af / tf / HBM_PEAK,
                'achieved': af / tf / 1e9, 'frac': af / tf / HBM_PEAK, 'traffic': None, 'levels': per,
                'kernel': 'mean-field layer, all %d decoder levels, forward (inference configuration)' % len(per)}
    return {'bound': 'hbm', 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'fwd_alg_bytes': af, 'fwd_us': tf * 1e6, 'fwd_frac': af / tf / HBM_PEAK,
            'fwd_bwd_alg_bytes': af + ab, 'fwd_bwd_us': (tf + tb) * 1e6, 'achieved': (af + ab) / (tf + tb) / 1e9,
            'frac': (af + ab) / (tf + tb) / HBM_PEAK, 'traffic': None, 'levels': per,
            'kernel': 'mean-field layer, all %d decoder levels, forward + backward (frac / achieved); fwd_frac = forward only' % len(per)}


def step_byte_model(B, N, K, T, n_cls, in_channels=6, widths=(32, 64, 128, 256, 512), ratio=(4, 4, 4, 4, 2)):
    """ALGORITHMIC bytes of ONE training step (fwd + loss + bwd + SGD) of PointConvBig at (B, N, K, T): every operator reads each of
    its operands once and writes each result once, fp32 values, int32 indices -- BatchNorm statistics, activations, residual joins and
    dropout counted as fused into the producing / consuming Linear (no pass of their own), nothing re-read.  The same accounting as
    SURVEY 8(d) uses for the mean-field kernel, extended to the network (DESIGN.md 6 has the table).  Returns (total, per-group dict).
      MLP(Ci -> Co) on m rows        fwd 4 m (Ci + Co)                bwd 4 m (2 Ci + 2 Co)   [gA, the saved output, X in; dX out]
      gather / pool, m_t <- m_s, C   fwd 4 K' m_t + 4 C (m_s + m_t)   bwd the same + reverse list 4 K' m_t + 4 m_s
      PointConv(d), m_t <- m_s       fwd m_t (4 K + 12 + 4 d) + m_s (12 + 4 d)    bwd 2 x fwd + reverse list
      mean field (H), m rows         fwd m (4 (K-1) + 4 H (2 T + 1))  bwd 2 x fwd + m (4 K + 4)"""
    m = [B * N]
    for r in ratio[:len(widths) - 1]:
        m.append(m[-1] // r)
    g = {}

    def add(name, f, b):
        a = g.setdefault(name, [0, 0])
        a[0] += f
        a[1] += b

    def mlp(name, rows, ci, co, extra_in=0):
        add(name, 4 * rows * (ci + co + extra_in), 4 * rows * (2 * ci + 2 * co + extra_in))

    def move(name, mt, ms, C, k):
        f = 4 * k * mt + 4 * C * (ms + mt)
        add(name, f, f + 4 * k * mt + 4 * ms)

    def pconv(mt, ms, d):
        f = mt * (4 * K + 12 + 4 * d) + ms * (12 + 4 * d)
        add('pointconv', f, 2 * f + 4 * K * mt + 4 * ms)
    cin = in_channels
    for lvl, w in enumerate(widths):
        for blk in range(2):
            strided = blk == 0 and lvl > 0
            ci = cin if blk == 0 else w
            ms = m[lvl - 1] if strided else m[lvl]
            mt = m[lvl]
            d = w // 4
            mlp('encoder_linear', ms, ci, d)                                     # lin_in
            pconv(mt, ms, d)
            mlp('encoder_linear', mt, d, w, extra_in=w)                          # lin_out + the residual it joins
            if ci != w:
                mlp('encoder_linear', ms, ci, w)                                 # shortcut
            if strided:
                move('pool_gather', mt, ms, w, K)                                # max-pool of the shortcut
        cin = w
    for lvl in range(len(widths) - 2, -1, -1):                                   # deconv4 .. deconv1
        U, P = widths[lvl + 1], widths[lvl]
        H, mc, mf = P // 4, m[lvl + 1], m[lvl]
        mlp('decoder_linear', mc, U, H)
        mlp('decoder_linear', mc, H, H)
        mlp('decoder_linear', mf, P, H)
        mlp('decoder_linear', mf, H, H)
        move('pool_gather', mf, mc, H, 1)                                        # nearest up-sampling of the unary term
        f = mf * (4 * (K - 1) + 4 * H * (2 * T + 1))
        add('mean_field', f, 2 * f + mf * (4 * K + 4))
        mlp('decoder_linear', mf, H, P)
        mlp('decoder_linear', mf, 2 * P, P)
    mlp('classifier_loss', m[0], widths[0], 4 * widths[0])
    mlp('classifier_loss', m[0], 4 * widths[0], n_cls)
    add('classifier_loss', m[0] * (4 * n_cls + 8), m[0] * (8 * n_cls + 8))
    return sum(a[0] + a[1] for a in g.values()), {k: {'fwd': v[0], 'bwd': v[1]} for k, v in g.items()}


def roofline_pointconv(data, dev, d=8):
    """Level-0 PointConv (d = 8: conv1_1 / conv1_2 of models/point_conv_big.py:116-117) in train mode, forward and
    forward + backward captured into hipGraphs (the op is four to ten launches; eagerly the host would be timed) and
    replayed, HIP-event timed.  Algorithmic bytes per target point, forward: 4 K (index row) + 12 (p_i) +
    4 d (output) + (12 + 4 d) (the source row, each read once) = 152 B at d = 8, K = 16; backward counted as 3 x that
    (the gradient row in, dx out, the rows again, plus the weight-MLP parameter sums)."""
    from crfconv_amd import ops
    from crfconv_amd.graph import table_of
    ms0 = data.multiscale[0]
    B, N, K = ms0.neighbor_idx.shape
    m = B * N
    tab = table_of(ms0.neighbor_idx, N)
    tab.reverse
    pos = ms0.pos.reshape(-1, 3).contiguous()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(m, d, generator=g).to(dev).requires_grad_()
    W1 = (0.5 * torch.randn(d, 3, generator=g)).to(dev).requires_grad_()
    W2 = (0.5 * torch.randn(d, d, generator=g)).to(dev).requires_grad_()
    bn1, bn2 = torch.nn.BatchNorm1d(d).to(dev), torch.nn.BatchNorm1d(d).to(dev)
    gout = torch.randn(m, d, generator=g).to(dev)
    moments = ops.relpos_moments(pos, pos, tab)

    def fwd():
        return ops.point_conv(x, pos, None, tab, W1, bn1, W2, bn2, True, moments=moments)

    def fwd_bwd():
        for t in (x, W1, W2, bn1.weight, bn1.bias, bn2.weight, bn2.bias):
            t.grad = None
        fwd().backward(gout)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gf, gfb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(gf):
        fwd()
    with torch.cuda.graph(gfb):
        fwd_bwd()
    tf, tf_lo = _event_time(gf.replay, per=5)
    tfb, _ = _event_time(gfb.replay, per=5)
    alg = m * (4 * K + 12 + 4 * d + 12 + 4 * d)
    pc_traffic, pc_note = _measured_traffic(TRAFFIC_PC, {'m': m, 'd': d, 'K': K})
    return {'bound': 'hbm', 'achieved': alg / tf / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': alg / tf / HBM_PEAK,
            'traffic': pc_traffic, 'traffic_source': pc_note,
            'kernel': 'PointConv level 0, d=%d, train mode (uvstats + combine; m=%d, K=%d), graph replay' % (d, m, K),
            'alg_bytes_per_launch': alg, 'avg_launch_us': tf * 1e6, 'min_launch_us': tf_lo * 1e6,
            'fwd_bwd_us': tfb * 1e6, 'bwd_frac_on_3x_bytes': 3 * alg / max(tfb - tf, 1e-9) / HBM_PEAK}


def _layer_summary(r):
    """The mean-field layer of one configuration against the HBM roofline on SURVEY 8(d)'s algorithmic bytes (C3 31.2 MB, C4 159.9 MB
    per GPU, C5 824.7 MB / 16 crops = 51.5 MB per crop, forward): every decoder level timed alone with HIP events, as `roofline_layer`."""
    if r is None:
        return None
    keep = ('bound', 'peak', 'unit', 'fwd_alg_bytes', 'fwd_us', 'fwd_frac', 'fwd_bwd_alg_bytes', 'fwd_bwd_us', 'achieved', 'frac', 'kernel')
    return {k: r[k] for k in keep if k in r}


