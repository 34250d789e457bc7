#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): M points/s, fwd+bwd(+SGD step) of PointConvBig with CRF decoders on
synthetic S3DIS-like clouds -- 4 clouds x 40960 points per GPU, K=16, 3 mean-field steps.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     the level-0 CRF mean-field forward (similarity + T steps) against the 8 TB/s HBM peak,
                  algorithmic bytes = m * (4 (K-1) + 4 H (2 T + 1))  (SURVEY.md 8(d)), timed with HIP events
  "cpu_baseline": the CPU oracle (oracle/crf_oracle.py, a port of the reference op sequence) on this host.
"""
import argparse
import json
import os
import sys
import time


def _self_launch():
    """`python bench.py --gpus N` WITHOUT a launcher's environment (no WORLD_SIZE): this process only starts the N ranks -- fresh
    children, one per GPU, with the torchrun variables (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT) --
    BEFORE it imports torch or crfconv_amd or touches the GPU (never a re-exec of a process that has), relays their output
    (rank 0 prints the JSON line), and exits non-zero if any rank fails.  N = 1 runs in-process as before unless --spawn asks
    for the one-rank process group (RCCL path with one rank).  Under torchrun (WORLD_SIZE set) nothing happens here."""
    import socket
    import subprocess
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--spawn', action='store_true')
    a, _ = ap.parse_known_args()
    if 'WORLD_SIZE' in os.environ or (a.gpus <= 1 and not a.spawn):
        return
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(max(1, a.gpus)):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(max(1, a.gpus)), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, t_fail = 0, None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
        if bad and t_fail is None:
            rc, t_fail = bad[0], time.time()                     # a rank died: the others hang in their next collective
        if t_fail is not None and time.time() - t_fail > 20:
            for p in procs:
                if p.poll() is None:
                    p.kill()                                     # exactly the PIDs started above
    for p in procs:
        if p.returncode != 0 and rc == 0:
            rc = p.returncode
    sys.exit(rc if rc is not None else 1)


if __name__ == '__main__':
    _self_launch()

import numpy as np        # noqa: E402
import crfconv_amd        # noqa: E402,F401  (first: picks the hardware-queue mapping for a launch under a process group, crfconv_amd/__init__.py)
import torch              # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12           # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
TRAFFIC_FWD, TRAFFIC_BWD = 'r6_meanfield_traffic.json', 'r6_meanfield_bwd_traffic.json'      # PMC passes, sha1-keyed to the kernel sources
TRAFFIC_STEP, TRAFFIC_PC = 'r5_step_traffic.json', 'r5_pointconv_traffic.json'
ROCPROF_MF = 'r6_meanfield_rocprof.json'      # rocprofv3 --kernel-trace average durations of the level-0 mean-field kernels, sha1-keyed
BOX = (8.0, 8.0, 3.0)
VOX = 0.04


def synth_cloud(seed, n):
    """One point per occupied 4 cm voxel of an 8 x 8 x 3 m box, jittered +-1 cm."""
    rng = np.random.default_rng(seed)
    dims = np.array([int(b / VOX) for b in BOX])
    flat = rng.choice(int(dims.prod()), size=n, replace=False)
    ijk = np.stack(np.unravel_index(flat, dims), -1).astype(np.float64)
    xyz = (ijk + 0.5) * VOX + rng.uniform(-0.01, 0.01, (n, 3))
    rgb = rng.uniform(0, 1, (n, 3))
    lab = rng.integers(1, 14, n)
    return xyz.astype(np.float32), rgb.astype(np.float32), lab.astype(np.int64)


def make_batch(rank, B, N, dev, gen, sort='morton'):
    import crfconv_amd
    clouds = [synth_cloud(rank * B + i, N) for i in range(B)]
    pos = torch.from_numpy(np.stack([c[0] for c in clouds])).to(dev)
    x = torch.cat([pos, torch.from_numpy(np.stack([c[1] for c in clouds])).to(dev)], -1)
    y = torch.from_numpy(np.stack([c[2] for c in clouds])).to(dev)
    t0 = time.perf_counter()
    data = crfconv_amd.multiscale_compute(pos, x=x, y=y, generator=gen, sort=sort)
    torch.cuda.synchronize()
    return data, time.perf_counter() - t0


def _meanfield_problem(data, dev, H, seed=1, level=0):
    from crfconv_amd.graph import table_of
    ms0 = data.multiscale[level]
    B, N, K = ms0.neighbor_idx.shape
    m = B * N
    tab = table_of(ms0.neighbor_idx, N)
    g = torch.Generator(device='cpu').manual_seed(seed)
    z = torch.randn(m, H, generator=g).to(dev)
    y = torch.randn(m, H, generator=g).to(dev)
    c = torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)
    C = c.t() @ c
    Q = torch.linalg.inv(torch.eye(H) + C)
    P = (C @ Q).to(dev).contiguous()
    return tab, m, K, z, y, Q.to(dev).contiguous(), P, g


def _event_time(launch, per=10, regions=20):
    """Average duration of one `launch()` on the current stream: HIP events around `per` consecutive launches (an event
    pair per launch adds ~3 us of record latency to a ~25 us region) give one average per region; returns the median of
    `regions` such averages and the smallest."""
    for _ in range(10):
        launch()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(regions)]
    for a, b in evs:
        a.record()
        for _ in range(per):
            launch()
        b.record()
    torch.cuda.synchronize()
    dur = np.array([a.elapsed_time(b) for a, b in evs]) * 1e-3 / per
    # every region is already the AVERAGE over `per` launches; across regions the median, so that one region hit by an unrelated
    # stall of the box (seen: a single 10 ms region among twenty ~25 us ones) does not decide the figure
    return float(np.median(dur)), float(dur.min())


def _measured_traffic(name, config):
    """HBM-side bytes per launch come from rocprofv3 PMC passes (they cannot be read live).  The committed measurement
    names the configuration AND the sha1 of the kernel source it was taken on: a changed kernel file or another shape
    reports null (with the reason) instead of a stale number."""
    import hashlib
    path = os.path.join(ROOT, 'profiles', name)
    try:
        rec = json.load(open(path))
        srcs = rec['source'] if isinstance(rec['source'], list) else [rec['source']]
        h = hashlib.sha1()
        for src in srcs:                      # sha1 over the concatenation of the kernel sources the measurement covers
            h.update(open(os.path.join(ROOT, src), 'rb').read())
        if h.hexdigest() != rec['source_sha1']:
            return None, 'stale: %s changed since the PMC passes of %s' % (', '.join(srcs), name)
        if any(rec['config'].get(k) != v for k, v in config.items()):
            return None, 'PMC passes of %s cover %s only' % (name, rec['config'])
        return rec['traffic_bytes_per_launch'], 'rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, %s' % rec.get('profile', name)
    except (OSError, KeyError, ValueError) as e:
        return None, 'no usable measurement (%s)' % type(e).__name__


def _rocprof_durations(config):
    """{'fwd_us', 'bwd_us'}: sums of the rocprofv3 --kernel-trace AVERAGE durations of the level-0 mean-field kernels (committed under
    profiles/ with the sha1 of the kernel sources, like the PMC traffic): the profiler's own clock beside the HIP-event one of this
    run -- `frac_rocprof` in the roofline objects.  (None, reason) when the sources changed since or the shape differs."""
    import hashlib
    path = os.path.join(ROOT, 'profiles', ROCPROF_MF)
    try:
        rec = json.load(open(path))
        h = hashlib.sha1()
        for src in rec['source']:
            h.update(open(os.path.join(ROOT, src), 'rb').read())
        if h.hexdigest() != rec['source_sha1']:
            return None, 'stale: %s changed since the kernel trace of %s' % (', '.join(rec['source']), ROCPROF_MF)
        if any(rec['config'].get(k) != v for k, v in config.items()):
            return None, 'the kernel trace of %s covers %s only' % (ROCPROF_MF, rec['config'])
        return rec, 'rocprofv3 --kernel-trace --stats, %s' % rec.get('profile', ROCPROF_MF)
    except (OSError, KeyError, ValueError) as e:
        return None, 'no usable kernel trace (%s)' % type(e).__name__


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


def copy_ceiling(dev, nbytes=1 << 28, iters=20):
    """Measured device-copy rate of this run (reported beside the 8 TB/s spec the fraction is taken of):
    read + written bytes of a 256 MiB float32 copy per second."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9


def _median_time(fn, warm, reps):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def cpu_share():
    """Host threads this process may really run: the cgroup CPU quota when there is one (a one-GPU box exposes all of the
    host's cores but grants a share of them; idle-spinning worker threads beyond the quota get the whole process throttled
    for the rest of the scheduler period -- seen as 70-90 ms stalls in the per-batch host code), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()                       # cgroup v2
    except (OSError, ValueError):
        try:
            quota = open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read().strip()               # cgroup v1
            period = open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read().strip()
        except OSError:
            return n
    if quota not in ('max', '-1'):
        n = min(n, max(1, int(quota) // int(period)))
    return n


def cpu_baseline(data, net, steps_T, labels, n_cls, dev):
    """SURVEY 8(d) CPU plan on this host's cores, bounded to ~40 s: (1) the oracle's fwd+bwd of PointConvBig on the SAME
    batch the GPU line is quoted on (all clouds: BatchNorm statistics span the batch), 3 warm-up + 5 timed, median;
    (2) the reference's OWN kNN (knn_.cxx cpp_knn_batch_omp, compiled unchanged into oracle/_ref) on the
    batch's level-0 self-query, K = 16; (3) the reference's own grid subsampling core on 2 M points, each 1 warm-up +
    3 timed, median, with the HIP kernels' times on the same inputs beside them."""
    from oracle import crf_oracle as O
    from oracle import native as onative
    import crfconv_amd
    from crfconv_amd.utils import cpp_subsampling, nearest_neighbors
    threads = torch.get_num_threads()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    prm = {k: v.requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    ms = [{k: getattr(l, k).cpu() for k in ('pos', 'neighbor_idx', 'sub_idx', 'up_idx')} for l in data.multiscale]
    x = data.x.cpu()
    y = labels.cpu()
    n = x.shape[0] * x.shape[1]

    def step():
        for v in prm.values():
            v.grad = None
        logits = O.pointconv_resnet(prm, x, ms, steps_T, True, True)
        O.training_loss(logits, y).backward()
    dt, ts = _median_time(step, 3, 5)
    out = {'value': n / dt / 1e6, 'unit': 'M points/s', 'cores': threads, 'kind': 'port',
           'sample': 'the whole batch (%d clouds x %d pts, K=16, T=%d): oracle/crf_oracle.py fwd+bwd, train mode, 3 warm-up + 5 timed '
                     '(%s s), median; os.cpu_count()=%d, torch intra-op threads actually used=%d (torch.get_num_threads(), sized '
                     'from the cgroup CPU quota), OMP_NUM_THREADS=%s'
                     % (x.shape[0], x.shape[1], steps_T, ', '.join('%.2f' % t for t in ts), os.cpu_count(), threads,
                        os.environ.get('OMP_NUM_THREADS', 'unset'))}
    # (2) kNN: the whole level-0 query of the batch
    pos = data.multiscale[0].pos
    pos_np = pos.cpu().numpy()
    Bc, Nc = pos_np.shape[:2]
    have_ref = onative.have_ref()
    knn_cpu = (lambda: onative.ref_knn_batch(pos_np, pos_np, 16, omp=True)) if have_ref else None
    if knn_cpu is not None:
        dt_knn, _ = _median_time(knn_cpu, 1, 3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nearest_neighbors.knn_batch_device(pos, pos, 16)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            nearest_neighbors.knn_batch_device(pos, pos, 16)
        e1.record()
        torch.cuda.synchronize()
        out['knn'] = {'cpu_queries_per_s': Bc * Nc / dt_knn, 'gpu_queries_per_s': Bc * Nc / (e0.elapsed_time(e1) * 1e-4),
                      'kind': 'reference', 'cores': min(Bc, os.cpu_count()),
                      'sample': "reference cpp_knn_batch_omp (OpenMP over the %d clouds, knn_.cxx:104-135), %d x %d self-queries, "
                                "K=16, median of 3: %.3f s; HIP kNN (grid build + query): %.3f ms" % (Bc, Bc, Nc, dt_knn, e0.elapsed_time(e1) / 10)}
    # (3) grid subsampling: 2 M points, 3 feature columns, 1 label column, 4 cm voxels
    rng = np.random.default_rng(5)
    npts = 2_000_000
    pts = (rng.random((npts, 3)) * np.array([20.0, 20.0, 5.0])).astype(np.float32)
    feats = rng.random((npts, 3)).astype(np.float32)
    cls = rng.integers(0, 13, (npts, 1)).astype(np.int32)
    grid_cpu = (lambda: onative.ref_grid_subsample(pts, feats, cls, 0.04)) if have_ref else (lambda: onative.oracle_grid_subsample(pts, feats, cls, 0.04))
    dt_grid, _ = _median_time(grid_cpu, 1, 3)
    dt_gpu, _ = _median_time(lambda: cpp_subsampling.compute(pts, features=feats, classes=cls, sampleDl=0.04), 1, 3)
    out['grid_subsample'] = {'cpu_points_per_s': npts / dt_grid, 'gpu_points_per_s_incl_pcie': npts / dt_gpu,
                             'kind': 'reference' if have_ref else 'port', 'cores': 1,
                             'sample': '%s, %d points + 3 features + 1 label, dl = 0.04, median of 3: %.3f s; '
                                       'cpp_subsampling.compute on the GPU incl. host<->device copies: %.3f s'
                                       % ('reference grid_subsampling.cpp core' if have_ref else 'oracle/grid_oracle.c', npts, dt_grid, dt_gpu)}
    return out


def other_configs(dev, rank=0):
    """Informational timings of the other BASELINE.json configs on ONE GPU (they are parity-test cases, tests/test_gpu_model.py, not bench
    lines; `--other-configs` adds this block to the JSON line): C3 one 122 880-point scan, T = 1, inference; C4 4 x 81 920-point clouds, T = 3,
    the training step (per-GPU share of the 8-GPU config); C5 one 65 536-point crop, K = 32, T = 5, inference.  hipGraph replays, HIP events."""
    import crfconv_amd
    from crfconv_amd import distributed as D
    from crfconv_amd import models, ops
    out = {}

    def batch(B, N, K, seed):
        clouds = [synth_cloud(seed + i, N) for i in range(B)]
        pos = torch.from_numpy(np.stack([c[0] for c in clouds])).to(dev)
        x = torch.cat([pos, torch.from_numpy(np.stack([c[1] for c in clouds])).to(dev)], -1)
        y = torch.from_numpy(np.stack([c[2] for c in clouds])).to(dev)
        return crfconv_amd.multiscale_compute(pos, x=x, y=y, kernel_size=(K,) * 5, generator=torch.Generator().manual_seed(seed), sort='morton')

    def replay_time(fn, warm=3):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warm):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return _event_time(g.replay, per=5)[0]
    for name, (B, N, K, T, C) in (('C3 KITTI-like scan, inference', (1, 122880, 16, 1, 19)), ('C5 Semantic3D crop, inference', (1, 65536, 32, 5, 8))):
        data = batch(B, N, K, 300 + N % 97)
        net = models.PointConvBig(6, C, use_crf=True, steps=T).to(dev).eval()
        with torch.no_grad():
            t = replay_time(lambda: net(data))
        out[name] = {'points': B * N, 'K': K, 'T': T, 'ms': t * 1e3, 'M_points_per_s': B * N / t / 1e6,
                     'roofline_meanfield_layer': _layer_summary(roofline_layer(data, dev, T, backward=False))}
        del net, data
    B, N, K, T, C = 4, 81920, 16, 3, 20
    data = batch(B, N, K, 400)
    net = models.PointConvBig(6, C, use_crf=True, steps=T).to(dev).train()
    bucket = D.FlatGradAllReduce(net)
    opt = crfconv_amd.optim.FlatSGD(bucket, lr=1e-2, momentum=0.95, weight_decay=1e-4)
    cw, unit = torch.ones(C, device=dev), torch.ones((), device=dev)

    def step():
        opt.zero_grad()
        loss = ops.training_loss(net(data), data.y, cw, ignore_index=-1)
        with ops.deferred_weight_grads(sink=bucket.view_of):
            loss.backward(unit)
        bucket.pack()
        opt.step()
    t = replay_time(step)
    out['C4 ScanNet-like batch (per-GPU share), training step'] = {'points': B * N, 'K': K, 'T': T, 'ms': t * 1e3, 'M_points_per_s': B * N / t / 1e6,
                                                                      'roofline_meanfield_layer': _layer_summary(roofline_layer(data, dev, T))}
    del net, data, bucket, opt
    out['pipelines'] = config_pipelines(dev)
    return out


def config_pipelines(dev):
    """Config 5 and config 3 as the PIPELINES north_star names them, eagerly as a user would call them:
    C5: one 1 048 576-point scene (60 x 60 x 15 m) -> PossibilitySampler (16 crops of 65 536 points) -> multiscale_compute(K = 32) ->
        PointConvBig(T = 5) eval -> VoteAccumulator.update -> project onto a 2 M-point raw cloud (trainval.py:170-203,
        datasets/semantic3d_dataset.py:423-460): whole-pipeline scene points/s and per-stage milliseconds;
    C3: one 122 880-point scan: multiscale_compute(K = 16) + PointConvBig(T = 1) eval."""
    import crfconv_amd
    from crfconv_amd import models
    from crfconv_amd.sampling import PossibilitySampler, VoteAccumulator, vote_scene
    from crfconv_amd.utils import nearest_neighbors
    out = {}
    g = torch.Generator().manual_seed(50)
    n_scene, n_crop, n_crops, K, T, C = 1 << 20, 65536, 16, 32, 5, 8
    pts = (torch.rand(n_scene, 3, generator=g) * torch.tensor([60.0, 60.0, 15.0])).to(dev)
    rgb = torch.rand(n_scene, 3, generator=g).to(dev)
    raw = (torch.rand(2 * n_scene, 3, generator=g) * torch.tensor([60.0, 60.0, 15.0])).to(dev)
    net = models.PointConvBig(6, C, use_crf=True, steps=T).to(dev).eval()

    def run(timings):
        smp = PossibilitySampler([pts], rgb=[rgb], num_points=n_crop, split='test', generator=torch.Generator().manual_seed(51))
        votes = VoteAccumulator([n_scene], C, device=dev)
        vote_scene(smp, net, votes, n_crops, kernel_size=(K,) * 5, generator=torch.Generator().manual_seed(52), timings=timings)
        return votes
    run(None)                                              # warm-up (allocator, lazily built tables, kernel modules)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    votes = run(None)
    torch.cuda.synchronize()
    t_loop = time.perf_counter() - t0
    stages = {}
    run(stages)                                            # the same again with a device synchronisation around every stage
    t0 = time.perf_counter()
    proj = nearest_neighbors.knn_batch_device(pts.unsqueeze(0), raw.unsqueeze(0), 1).reshape(-1)      # offline in the reference (sklearn KDTree.query)
    torch.cuda.synchronize()
    t_proj_idx = time.perf_counter() - t0
    t0 = time.perf_counter()
    labels = votes.project(0, proj)
    torch.cuda.synchronize()
    t_project = time.perf_counter() - t0
    votes.check()
    covered = float((votes.test_probs[0].sum(1) > 0).float().mean())
    out['C5 Semantic3D-like scene, tiled inference'] = {
        'scene_points': n_scene, 'crops': n_crops, 'crop_points': n_crop, 'K': K, 'T': T, 'classes': C,
        'loop_ms': t_loop * 1e3, 'ms_per_crop': t_loop * 1e3 / n_crops, 'crop_points_per_s_M': n_crops * n_crop / t_loop / 1e6,
        'scene_points_per_s_M': n_scene / (t_loop + t_project) / 1e6,
        'stage_ms_per_crop_synchronised': {k: v / n_crops for k, v in stages.items()},
        'project_ms': t_project * 1e3, 'raw_points_projected': int(raw.shape[0]), 'projection_index_ms_offline': t_proj_idx * 1e3,
        'scene_fraction_voted': covered, 'labels_histogram': torch.bincount(labels.long(), minlength=C + 1).tolist(),
        'what': 'sampler -> multiscale_compute(K=32) -> PointConvBig(T=5).eval() -> votes for 16 crops (eager, B = 1 per crop as the sampler yields '
                'them), then the arg-max re-projection onto a raw cloud; loop_ms is wall time without per-stage synchronisation, the stage '
                'figures come from a second run that synchronises around every stage; scene_points_per_s = scene points / (loop + projection)'}
    del net, votes
    # C3: one scan, collate + network
    N3, C3 = 122880, 19
    cl = synth_cloud(310, N3)
    pos3 = torch.from_numpy(cl[0]).to(dev).unsqueeze(0)
    x3 = torch.cat([pos3, torch.from_numpy(cl[1]).to(dev).unsqueeze(0)], -1)
    net3 = models.PointConvBig(6, C3, use_crf=True, steps=1).to(dev).eval()

    def scan():
        d = crfconv_amd.multiscale_compute(pos3, x=x3, kernel_size=(16,) * 5, generator=torch.Generator().manual_seed(5), sort='morton')
        with torch.no_grad():
            return net3(d)
    for _ in range(3):
        scan()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        scan()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t3 = float(np.median(ts))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        d3 = crfconv_amd.multiscale_compute(pos3, x=x3, kernel_size=(16,) * 5, generator=torch.Generator().manual_seed(5), sort='morton')
    torch.cuda.synchronize()
    t_col = (time.perf_counter() - t0) / 5
    out['C3 KITTI-like scan, collate + inference'] = {'points': N3, 'K': 16, 'T': 1, 'ms': t3 * 1e3, 'M_points_per_s': N3 / t3 / 1e6,
                                                       'collate_ms': t_col * 1e3,
                                                       'what': 'eager multiscale_compute (Morton sort, kNN at five scales, subsets, up-indices) + PointConvBig(T=1).eval() '
                                                               'forward per scan, wall clock, median of 10'}
    return out


def _parity_report():
    """ABSOLUTE logit error per BASELINE config (north_star: per-point logits within 1e-4 fp32): written by the -m gpu suite
    (tests/test_gpu_model.py::_eval_net_vs_oracle under CRFCONV_PARITY_RECORD, whole PointConvBig in eval mode at each config's full
    size against the float64 run of the CPU oracle) and committed as tests/golden/parity_report.json -- max |logit - oracle|, the same
    normalised by max(1, max |logit|), rows beyond 1e-4 absolute, and the float32 ORACLE's own distance from float64 beside them."""
    try:
        rec = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'parity_report.json')))
    except (OSError, ValueError) as e:
        return {'error': 'tests/golden/parity_report.json: %s' % type(e).__name__}
    rec['source'] = 'tests/golden/parity_report.json (recorded by `CRFCONV_PARITY_RECORD=... pytest tests -m gpu` on an MI355X; the tests assert these bounds in every run)'
    return rec


def _layer_summary(r):
    """The mean-field layer of one configuration against the HBM roofline on SURVEY 8(d)'s algorithmic bytes (C3 31.2 MB, C4 159.9 MB
    per GPU, C5 824.7 MB / 16 crops = 51.5 MB per crop, forward): every decoder level timed alone with HIP events, as `roofline_layer`."""
    if r is None:
        return None
    keep = ('bound', 'peak', 'unit', 'fwd_alg_bytes', 'fwd_us', 'fwd_frac', 'fwd_bwd_alg_bytes', 'fwd_bwd_us', 'achieved', 'frac', 'kernel')
    return {k: r[k] for k in keep if k in r}


def reference_loop(net, data, cw, steps):
    """The reference's training step, verbatim (trainval.py:99-106): optimizer.zero_grad(); y_pred = model(data);
    y = data.y.reshape(-1) - 1; loss = F.cross_entropy(y_pred, y, weight, ignore_index=-1); loss.backward(); optimizer.step()
    with torch.optim.SGD(lr=1e-2, momentum=0.95, weight_decay=1e-4) -- (a) eagerly, as a user who only swaps the import gets it,
    (b) through crfconv_amd.train.CapturedStep (the same five lines as one hipGraph replay)."""
    import torch.nn.functional as F
    from crfconv_amd.train import CapturedStep
    opt = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.95, weight_decay=1e-4)

    def loss_fn(y_pred, d):
        return F.cross_entropy(y_pred, d.y.reshape(-1) - 1, weight=cw, ignore_index=-1)

    def one():
        opt.zero_grad()
        loss = loss_fn(net(data), data)
        loss.backward()
        opt.step()
        return loss.detach()      # (a live loss would keep this iteration's autograd nodes -- AccumulateGrad included -- alive)
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps * 1e3
    out = {'eager_ms_per_step': eager, 'eager_final_loss': float(loss),
           'what': 'trainval.py:99-106 unchanged (zero_grad, model(data), F.cross_entropy(weight, ignore_index=-1), backward, '
                   'torch.optim.SGD.step), %d timed steps after 3 warm-up' % steps}
    for key, defer in (('captured_as_written_ms_per_step', False), ('captured_ms_per_step', True)):
        # CapturedStep's default batches the ~150 weight-gradient launches of the backward (ops.deferred_weight_grads inside the
        # capture; the caller's five lines are untouched); "as written" = the backward exactly as autograd issues it
        step = CapturedStep(net, opt, loss_fn, data, defer_weight_grads=defer)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        out[key] = (time.perf_counter() - t0) / steps * 1e3
        del step
    out['captured_final_loss'] = float(loss)
    # (c) the five lines verbatim again, on the model wrapped ONCE in crfconv_amd.train.GraphedModel: model(data) and loss.backward()
    # are one hipGraph replay each, F.cross_entropy and torch.optim.SGD.step stay the caller's eager code
    from crfconv_amd.train import GraphedModel
    bare, net = net, GraphedModel(net)
    opt = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.95, weight_decay=1e-4)
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    out['graphed_module_ms_per_step'] = (time.perf_counter() - t0) / steps * 1e3
    out['graphed_module_final_loss'] = float(loss)
    # (d) the loop the reference actually runs: `for data in train_loader:` (trainval.py:96) hands over a NEW collated batch every
    # step.  Here the collate is crfconv_amd.multiscale_compute on the device (the reference's runs in the DataLoader on the host,
    # datasets/semantic3d_dataset.py:501-534).  Four different raw batches take turns.
    import crfconv_amd
    from crfconv_amd.data import CollateGraph
    B, N = data.x.shape[:2]
    pool = []
    for r in range(4):
        clouds = [synth_cloud(9000 + 10 * r + i, N) for i in range(B)]
        pos_r = torch.from_numpy(np.stack([c[0] for c in clouds])).to(data.x.device)
        pool.append((pos_r, torch.cat([pos_r, torch.from_numpy(np.stack([c[1] for c in clouds])).to(pos_r.device)], -1),
                     torch.from_numpy(np.stack([c[2] for c in clouds])).to(pos_r.device)))
    gen = torch.Generator().manual_seed(4242)

    def one_on(d):
        opt.zero_grad()
        loss = loss_fn(net(d), d)
        loss.backward()
        opt.step()
        return loss.detach()

    def timed(body, n):
        for i in range(3):
            body(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            loss = body(3 + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, float(loss)
    # (d1) GraphedModel: the next batch's collate as a graph on a side stream while this step trains, its load into the captured batch as a
    # second graph between two steps (CollateGraph.collate / .load)
    cg = CollateGraph(net.static, generator=gen)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    staged, loaded = torch.cuda.Event(), torch.cuda.Event()
    cg.collate(*pool[0])                      # captures both graphs, stages batch 0
    staged.record()

    def graphed_fresh(i):
        main.wait_event(staged)
        cg.load()                             # captured batch <- staged batch (copy + in-place table refresh: one replay)
        loaded.record()
        side.wait_event(loaded)
        with torch.cuda.stream(side):
            cg.collate(*pool[(i + 1) % 4])    # kNN etc. of the NEXT batch beside this step
            staged.record()
        return one_on(net.static)
    out['fresh_graphed_ms_per_step'], out['fresh_graphed_final_loss'] = timed(graphed_fresh, steps)
    torch.cuda.synchronize()
    # (d2) the same, everything on one stream (collate graph, then the step)
    cg1 = CollateGraph(net.static, generator=gen)

    def graphed_fresh_serial(i):
        cg1.run(*pool[i % 4])
        return one_on(net.static)
    out['fresh_graphed_one_stream_ms_per_step'], _ = timed(graphed_fresh_serial, steps)
    net = bare
    opt = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.95, weight_decay=1e-4)
    # (d3) nothing wrapped, nothing captured: eager collate + the five lines on the bare model
    def eager_fresh(i):
        pos_r, x_r, y_r = pool[i % 4]
        return one_on(crfconv_amd.multiscale_compute(pos_r, x=x_r, y=y_r, generator=gen, sort='morton'))
    out['fresh_eager_ms_per_step'], out['fresh_eager_final_loss'] = timed(eager_fresh, steps)
    out['fresh_what'] = ('a NEW batch every step (4 x 40 960-point clouds, device collate = the reference\'s _multiscale_compute_fn): fresh_eager = '
                         'crfconv_amd.multiscale_compute + the unchanged five lines on the bare model; fresh_graphed = train.GraphedModel + '
                         'data.CollateGraph.collate (side stream, beside the step) / .load (between steps); ..._one_stream = CollateGraph.run then the step')
    return out


def rehearse(args):
    """`bench.py --gpus N --rehearse`: the rank plumbing of the N-GPU run WITHOUT a GPU -- every rank joins a gloo group on the CPU
    (no HIP call: a box allows six processes on its card, an 8-GPU node is not ours to launch on), replicas are broadcast from rank
    0, K steps of [forward, backward, pack + failure flag, ONE flat all-reduce, update] run on a small CPU network through the SAME
    FlatGradAllReduce / broadcast_parameters / barrier + MAX-over-ranks timing code the real run uses, and rank 0 prints one JSON
    line.  What it catches before a multi-GPU lease is spent on it: port / environment propagation of the self-launch, a rank that
    never joins, unequal replicas, the straggler handling, thread oversubscription (N ranks x host threads).  It measures NOTHING
    about the kernels or RCCL: no scaling number exists until the driver's SCALE_r*.json has N > 1."""
    from crfconv_amd import distributed as D
    rank, world, local = D.init_from_env(use_gpu=False)
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher environment says WORLD_SIZE=%d' % (args.gpus, world))
    torch.set_num_threads(1)
    torch.manual_seed(100 + rank)                            # different initial replicas: the broadcast must make them equal
    net = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.LeakyReLU(0.1), torch.nn.Linear(32, 13))
    D.broadcast_parameters(net)
    bucket = D.FlatGradAllReduce(net)
    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()
    g = torch.Generator().manual_seed(7 + rank)
    x, y = torch.randn(args.points, 6, generator=g), torch.randint(0, 13, (args.points,), generator=g)

    def step():
        bucket.zero()
        loss = torch.nn.functional.cross_entropy(net(x), y)
        loss.backward()
        bucket.pack()
        bucket.publish_guard()
        bucket.allreduce_packed()
        with torch.no_grad():
            for p, gview in zip(bucket.params, bucket.views):
                p.add_(gview, alpha=-1e-2 / world)
        return loss.detach()
    for _ in range(args.warmup):
        step()
    if grouped:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if grouped:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    spread = torch.zeros(1, dtype=torch.float64)
    if grouped:
        tt = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
        lo, hi = flat.clone(), flat.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        spread[0] = float((hi - lo).abs().max())
    if rank == 0:
        print(json.dumps({'rehearsal': True, 'metric': 'rank plumbing only (CPU, gloo): NOT a benchmark', 'world': world, 'n_gpus': 0,
                          'dist_backend': torch.distributed.get_backend() if grouped else None, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': dt / max(args.steps, 1) * 1e3, 'replica_spread': float(spread[0]), 'guard_slot': float(bucket.guard),
                          'final_loss': float(loss), 'points_per_rank': args.points}))
    if grouped:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--spawn', action='store_true', help='start the ranks as child processes even for --gpus 1 (one-rank RCCL group)')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=4, help='clouds per GPU')
    ap.add_argument('--points', type=int, default=40960)
    ap.add_argument('--crf-steps', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--pipe-gate', default='backward', choices=['off', 'forward', 'backward'],
                    help="the pipelined loop's device-side gate: the collate graph starts when the training step's forward ('forward') / backward ('backward') reaches its coarse levels, or wherever the launches fall ('off')")
    ap.add_argument('--mfma-min-rows', type=int, default=0, help='EXPERIMENT knob (A/B runs only): rows from which the row-streaming Linear forms take over from the tiled ones (ops.state.mfma_min_rows; 0 = the shipped 12288)')
    ap.add_argument('--rehearse', action='store_true', help='rank plumbing only, on the CPU over gloo (no GPU call, no benchmark): see rehearse()')
    ap.add_argument('--other-configs', action='store_true', help='(default since round 5; kept for old command lines)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the block that times BASELINE configs 3, 4 (per-GPU share) and 5 on this GPU')
    ap.add_argument('--graph', type=int, default=1, help='capture the training step into a hipGraph (1) or run eagerly (0)')
    ap.add_argument('--sort', default='morton', choices=['morton', 'none'],
                    help="point order emitted by the device collate (kernels are order-agnostic)")
    args = ap.parse_args()
    if args.rehearse:
        return rehearse(args)

    import crfconv_amd
    from crfconv_amd import distributed as D
    from crfconv_amd import models, ops

    if args.mfma_min_rows > 0:
        ops.state.mfma_min_rows = args.mfma_min_rows
    rank, world, local = D.init_from_env()
    torch.set_num_threads(max(1, min(torch.get_num_threads(), cpu_share() // max(1, world), 16)))     # 16 = the share per GPU of the pool's boxes
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher environment says WORLD_SIZE=%d' % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    B, N, T, n_cls = args.batch, args.points, args.crf_steps, 13

    gen = torch.Generator().manual_seed(1234 + rank)
    make_batch(rank, B, N, dev, gen, args.sort)          # warm-up of the collate kernels
    data, t_pre = make_batch(rank, B, N, dev, gen, args.sort)
    torch.manual_seed(0)
    net = models.PointConvBig(6, n_cls, use_crf=True, steps=T).to(dev).train()
    D.broadcast_parameters(net)
    bucket = D.FlatGradAllReduce(net)
    # torch.optim.SGD(lr, momentum=0.95, weight_decay=1e-4) of trainval.py:69-72 as one launch over flat parameters
    opt = crfconv_amd.optim.FlatSGD(bucket, lr=1e-2, momentum=0.95, weight_decay=1e-4, grad_scale=1.0 / world)   # mean of the ranks' gradients inside the update
    cw = torch.ones(n_cls, device=dev)
    unit = torch.ones((), device=dev)                        # d loss / d loss, made once (loss.backward() fills a new one per call)

    # One training step = [A] zero grads, forward, weighted CE, backward, grads packed into ONE flat fp32 bucket
    #                     [C] all-reduce of that bucket over RCCL (only when world > 1; eager, never captured)
    #                     [B] SGD(momentum, weight decay) step on views of the bucket.
    # A and B are each captured into a hipGraph (the step issues >1000 small launches; replaying them removes
    # the Python host from the critical path).  The collective stays outside the graphs on purpose.
    def part_a(d=None):
        d = data if d is None else d
        opt.zero_grad()
        logits = net(d)
        loss = ops.training_loss(logits, d.y, cw, ignore_index=-1)             # trainval.py:101-104, fused kernel
        with ops.deferred_weight_grads(sink=bucket.view_of):     # one batched launch finishes all 74 dW / db reductions, into the bucket
            loss.backward(unit)
        bucket.pack()                                     # one batched copy into the flat bucket; .grad -> bucket views
        if world > 1:
            bucket.publish_guard()                        # this rank's barrier-failure flag into the slot the all-reduce sums with the gradients
        return loss.detach()

    def part_b():
        opt.step()                                           # reads bucket.flat, updates the flat parameter vector

    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()

    def collective():
        if grouped:
            bucket.allreduce_packed()                     # ONE all-reduce (sum) of gradients + failure flag; the 1 / world factor sits in the SGD kernel

    def step():
        loss = part_a()
        collective()
        part_b()
        return loss

    def barrier():
        torch.cuda.synchronize()
        if grouped:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graph_note = 'eager'
    if args.graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, capture_error_mode='thread_local'):       # RCCL's watchdog thread keeps polling events
                static_loss = part_a()
            with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode='thread_local'):
                part_b()

            def step():                                   # noqa: F811
                ga.replay()
                collective()
                gb.replay()
                return static_loss
            graph_note = 'hipGraph replay (fwd+loss+bwd | eager RCCL all-reduce | SGD step)'
        except Exception as e:                            # capture not possible: stay eager, say so
            import traceback
            traceback.print_exc()
            torch.cuda.synchronize()
            graph_note = 'eager (graph capture failed: %s)' % str(e).splitlines()[0][:120]
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    rank_ms = {'min': dt / args.steps * 1e3, 'max': dt / args.steps * 1e3}
    if grouped:                           # the contract's figure = MAX over ranks; min and max are both on the line so that a straggler shows
        tt = torch.tensor([dt, -dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt[0].item())
        rank_ms = {'min': -float(tt[1].item()) / args.steps * 1e3, 'max': dt / args.steps * 1e3}
    ms_per_step = dt / args.steps * 1e3
    value = world * B * N / (dt / args.steps) / 1e6

    # ---- per-batch costs that the timed loop (ONE resident batch, cached tables) does not pay, and the proof that the
    # captured step trains on a FRESH batch.  A new batch is collated (kNN etc. on the GPU), copied into the static
    # buffers the graph reads (MultiScaleData.load_: narrowed tables, reverse CSRs, rel-pos moments refreshed in place)
    # and the graph replayed; the same step is then repeated eagerly from the same weights on an independent collate of
    # the same clouds: loss and updated parameters must agree.
    def new_batch(seed0):
        clouds = [synth_cloud(seed0 + rank * B + i, N) for i in range(B)]
        pos_n = torch.from_numpy(np.stack([c[0] for c in clouds])).to(dev)
        x_n = torch.cat([pos_n, torch.from_numpy(np.stack([c[1] for c in clouds])).to(dev)], -1)
        y_n = torch.from_numpy(np.stack([c[2] for c in clouds])).to(dev)
        g2 = torch.Generator().manual_seed(seed0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = crfconv_amd.multiscale_compute(pos_n, x=x_n, y=y_n, generator=g2, sort=args.sort)
        torch.cuda.synchronize()
        return d, time.perf_counter() - t0

    # MultiScaleData.load_ of a fresh batch into the static buffers: device time by HIP events, host time of the call, and the wall time
    # per batch of a loop that loads batch after batch with ONE synchronisation at its end (what a training loop pays: it does not
    # synchronise per batch; load_(defer_check=True) makes no host synchronisation of its own).  A synchronising call right behind a
    # single load_ is reported too: on this stack a device synchronisation behind a short burst of work returns after 10 or 20 ms
    # (timer-tick granularity of the blocking wait) -- that, not the refresh, was round 5's "22 ms per fresh batch".
    fresh_pool, t_collate = [], []
    for rep in range(4):
        nd, tc = new_batch(5000 + 100 * rep)
        fresh_pool.append(nd)
        t_collate.append(tc)
    t_load_host, t_load_dev, t_load_sync = [], [], []
    for rep in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        data.load_(fresh_pool[rep % 4], defer_check=True)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t_load_sync.append(time.perf_counter() - t0)
        t_load_host.append(t1 - t0)
        t_load_dev.append(e0.elapsed_time(e1) * 1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(16):
        data.load_(fresh_pool[rep % 4], defer_check=True)
    torch.cuda.synchronize()
    t_load = [(time.perf_counter() - t0) / 16]
    # the same per-batch work as ONE hipGraph replay (data.CollateGraph: collate + in-place refresh of the static batch)
    from crfconv_amd.data import CollateGraph
    t_graph, cg_parts = None, None
    try:
        cg = CollateGraph(data, generator=torch.Generator().manual_seed(99 + rank))
        raw = [synth_cloud(7000 + rank * B + i, N) for i in range(B)]
        pos_r = torch.from_numpy(np.stack([c[0] for c in raw])).to(dev)
        x_r = torch.cat([pos_r, torch.from_numpy(np.stack([c[1] for c in raw])).to(dev)], -1)
        y_r = torch.from_numpy(np.stack([c[2] for c in raw])).to(dev)
        cg.run(pos_r, x_r, y_r)                           # captures
        torch.cuda.synchronize()
        ts = []
        for rep in range(7):
            t0 = time.perf_counter()
            cg.run(pos_r, x_r, y_r)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t_graph = float(np.median(ts))

        def part(fn):
            torch.cuda.synchronize()
            v = []
            for _ in range(5):
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                v.append(time.perf_counter() - t0)
            return float(np.median(v)) * 1e3
        from crfconv_amd.data import morton_order
        cg_parts = {'host_work_ms': 0.0 if cg.device_draw else part(cg._draw),      # device_draw: subsets drawn inside the graph
                    'morton_argsort_alone_ms': part(lambda: morton_order(cg.pos, out=cg.order)),     # (inside the graph too)
                    'graph_replay_ms': part(cg.graph.replay), 'all_runs_ms': [round(t * 1e3, 2) for t in ts]}
    except Exception as e:                                # capture not possible: the eager figures above stand
        import traceback
        traceback.print_exc()
        torch.cuda.synchronize()
    nd, _ = new_batch(5200)
    data.load_(nd)                                        # the batch of seed 5200 for the comparison below
    torch.cuda.synchronize()
    buffers = {k: v.clone() for k, v in net.named_buffers()}
    flat0, mom0 = opt.flat.clone(), opt.buf.clone()
    loss_graph = float(step())                          # replay on the batch loaded last (seed 5200)
    torch.cuda.synchronize()
    flat_graph = opt.flat.clone()
    with torch.no_grad():
        opt.flat.copy_(flat0)
        opt.buf.copy_(mom0)
        for k, v in net.named_buffers():
            v.copy_(buffers[k])
    eager_data, _ = new_batch(5200)
    from crfconv_amd.graph import table_of
    tables_equal = True                                 # the refreshed static tables vs tables built from scratch
    for lvl_s, lvl_e in zip(data.multiscale, eager_data.multiscale):
        for name in ('neighbor_idx', 'sub_idx', 'up_idx'):
            a_, b_ = getattr(lvl_s, name, None), getattr(lvl_e, name, None)
            if a_ is None or not getattr(a_, '_crf_tables', None):
                continue
            for key, (tab_s, _) in a_._crf_tables.items():
                tab_e = table_of(b_, key[0])
                tables_equal &= bool(torch.equal(tab_s.idx32, tab_e.idx32))
                if tab_s._rev is not None:
                    tables_equal &= all(bool(torch.equal(u, v)) for u, v in zip(tab_s.reverse, tab_e.reverse))
    loss_eager = float(part_a(eager_data))
    collective()
    part_b()
    torch.cuda.synchronize()
    fresh = {'static_tables_equal_fresh_tables': tables_equal, 'loss_graph_replay': loss_graph,
             'loss_eager_same_weights': loss_eager,
             'max_param_diff_after_step': float((opt.flat - flat_graph).abs().max()),
             'collate_ms_per_batch': float(np.median(t_collate)) * 1e3,
             'table_refresh_ms_per_batch': float(np.median(t_load)) * 1e3, 'table_refresh_host_ms_per_batch': float(np.median(t_load_host)) * 1e3,
             'table_refresh_device_ms_per_batch': float(np.median(t_load_dev)) * 1e3,
             'table_refresh_plus_device_synchronize_ms': [round(v * 1e3, 2) for v in t_load_sync],
             'note': 'fresh batch -> MultiScaleData.load_ into the static buffers -> hipGraph replay, against an eager step '
                     'from the same weights and BatchNorm counters on an independent collate of the same clouds; the classifier\'s dropout mask '
                     'is keyed on (seed, step counter, element), so both runs draw the same mask and the losses agree; the refreshed '
                     'tables are compared bit for bit'}
    fresh['collate_plus_refresh_graph_ms_per_batch'] = None if t_graph is None else t_graph * 1e3
    fresh['collate_graph_parts'] = cg_parts if t_graph is not None else None
    per_batch_ms = ms_per_step + (t_graph * 1e3 if t_graph is not None else
                                  fresh['collate_ms_per_batch'] + fresh['table_refresh_ms_per_batch'])

    # ---- the training loop as it would run on a stream of fresh batches: two static batches, the collate graph of batch
    # i+1 on a side stream while the captured step of batch i trains (data.CollatePipeline).  EVERY iteration collates a
    # batch (host subset draw + Morton argsort + kNN at 5 scales + table / reverse-CSR / moment refresh) and trains on it.
    pipe_ms = None
    pipe_gate = None
    if t_graph is not None and graph_note.startswith('hipGraph'):
        try:
            from crfconv_amd.data import CollatePipeline
            data2, _ = make_batch(rank, B, N, dev, gen, args.sort)
            part_a(data2)                                 # builds the second batch's tables
            torch.cuda.synchronize()
            ga2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga2, pool=ga.pool(), capture_error_mode='thread_local'):
                part_a(data2)
            gas = (ga, ga2)
            gate = args.pipe_gate != 'off'
            raws = [(pos_r, x_r, y_r)]
            raw = [synth_cloud(9000 + rank * B + i, N) for i in range(B)]
            pos_q = torch.from_numpy(np.stack([c[0] for c in raw])).to(dev)
            raws.append((pos_q, torch.cat([pos_q, torch.from_numpy(np.stack([c[1] for c in raw])).to(dev)], -1),
                         torch.from_numpy(np.stack([c[2] for c in raw])).to(dev)))
            prio = os.environ.get('CRFCONV_BENCH_COLLATE_PRIORITY')     # A/B: side-stream priority (default: the lowest)
            pipe = CollatePipeline([data, data2], generator=torch.Generator().manual_seed(77 + rank),
                                   priority=None if prio is None else int(prio), gate=gate)
            pipe.submit(0, *raws[0])
            pipe.submit(1, *raws[1])                      # both collate graphs captured
            if gate:
                # the training graphs of the pipelined loop carry ONE more launch: a mark where the forward reaches its coarse levels
                # (PointConvBig.phase_hook); the collate graph on the side stream starts behind a bounded wait for it, so its kernels
                # fall beside the step's coarse-level launches (most of the chip idle) instead of beside its fine-level ones
                net.phase_hook = pipe.mark_on('coarse' if args.pipe_gate == 'forward' else 'coarse_backward')
                gated = []
                for d in (data, data2):
                    gg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gg, pool=ga.pool(), capture_error_mode='thread_local'):
                        part_a(d)
                    gated.append(gg)
                net.phase_hook = None
                gas = tuple(gated)
                pipe.enable_gate(True)

            def run_pipe(n, i0):
                for i in range(i0, i0 + n):
                    slot = i % 2
                    pipe.acquire(slot)
                    gas[slot].replay()
                    # next batch: its host part (subset draw, argsort launches) now runs while the step above computes; the
                    # side stream waits only for the release of the slot it overwrites (the clouds were uploaded long ago)
                    pipe.submit(1 - slot, *raws[(i + 1) % 2], wait_current=False)
                    collective()
                    gb.replay()
                    pipe.release(slot)
            pipe.submit(0, *raws[0])
            sprio = os.environ.get('CRFCONV_BENCH_STEP_PRIORITY')       # A/B: the training graphs on a stream of this priority
            if sprio is not None:
                hp = torch.cuda.Stream(priority=int(sprio))
                hp.wait_stream(torch.cuda.current_stream())
                ctx = torch.cuda.stream(hp)
            else:
                import contextlib
                ctx = contextlib.nullcontext()
            with ctx:
                run_pipe(4, 0)
                barrier()
                t0 = time.perf_counter()
                run_pipe(args.steps, 4)
                barrier()
                pipe_ms = (time.perf_counter() - t0) / args.steps * 1e3
            if sprio is not None:
                torch.cuda.current_stream().wait_stream(hp)
            if grouped:
                tt = torch.tensor([pipe_ms], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                pipe_ms = float(tt.item())
            pipe_gate = {'mode': args.pipe_gate, 'timeouts': pipe.gate_timeouts(), 'still_on': pipe.gate_is_on()}
            pipe.enable_gate(False)
        except Exception:
            import traceback
            traceback.print_exc()
            torch.cuda.synchronize()
            pipe_ms = None

    # ---- the all-reduce alone (HIP events around 20 calls on the launching stream), and what the group looks like
    allreduce_us, allreduce_host_us, backend_name = None, None, None
    if grouped:
        backend_name = torch.distributed.get_backend()
        for _ in range(5):
            collective()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            collective()
        e1.record()
        torch.cuda.synchronize()
        allreduce_us = e0.elapsed_time(e1) / 20 * 1e3
        t0 = time.perf_counter()          # ... and what ISSUING it costs the host (the call sits between two graph replays)
        for _ in range(20):
            collective()
        allreduce_host_us = (time.perf_counter() - t0) / 20 * 1e6
        torch.cuda.synchronize()

    # ---- what a caller of the UNCHANGED reference loop gets (trainval.py:99-106), on the same model and batch: no graph, no
    # FlatSGD, no deferred weight gradients, no fused loss -- torch.optim.SGD and F.cross_entropy as the reference writes them;
    # and the same loop handed to crfconv_amd.train.CapturedStep (one hipGraph replay per step, still torch.optim.SGD).
    ref_loop = None
    if rank == 0:
        try:
            ref_loop = reference_loop(net, data, cw, args.steps)
        except Exception as e:
            import traceback
            traceback.print_exc()
            torch.cuda.synchronize()
            ref_loop = {'error': str(e).splitlines()[0][:200]}

    if rank == 0:
        out = {
            'metric': 'M points/sec fwd+bwd, S3DIS 40960-pt cloud, K=16, 3 CRF iters; mIoU parity',
            'value': value, 'unit': 'M points/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: S3DIS-like synthetic clouds (one point per 4 cm voxel of an '
                                   '8x8x3 m box), %d clouds x %d pts per GPU, K=16, ratios [4,4,4,4,2], '
                                   'PointConvBig(in=6, classes=13, use_crf, steps=%d), train mode: fwd + weighted CE + '
                                   'bwd + flat-bucket grad all-reduce + SGD(momentum) step; tables resident in HBM; point order from '
                                   'the device collate: %s' % (B, N, T, args.sort),
                       'global_batch': world * B, 'points_per_cloud': N, 'parallelism': 'dp%d (batch-sharded)' % world},
            'final_loss': float(loss),
            'preprocess_ms_per_batch': fresh['collate_ms_per_batch'],
            'table_refresh_ms_per_batch': fresh['table_refresh_ms_per_batch'],
            'preprocess_plus_refresh_graph_ms_per_batch': fresh['collate_plus_refresh_graph_ms_per_batch'],
            'value_incl_preprocessing_serial': world * B * N / (per_batch_ms * 1e-3) / 1e6,
            'pipelined_ms_per_batch': pipe_ms, 'pipelined_gate': pipe_gate,
            'value_incl_preprocessing': world * B * N / ((pipe_ms or per_batch_ms) * 1e-3) / 1e6,
            'fresh_batch_replay': fresh,
            'launch_mode': graph_note,
            'rccl_ranks_seen': torch.distributed.get_world_size() if grouped else 1,
            'dist_backend': backend_name, 'allreduce_us': allreduce_us, 'allreduce_host_issue_us': allreduce_host_us,
            'rank_ms_per_step': rank_ms, 'gpu_max_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES'),
            'trainval_eager_ms_per_step': None if ref_loop is None else ref_loop.get('eager_ms_per_step'),
            'trainval_captured_ms_per_step': None if ref_loop is None else ref_loop.get('captured_ms_per_step'),
            'trainval_graphed_module_ms_per_step': None if ref_loop is None else ref_loop.get('graphed_module_ms_per_step'),
            'trainval_captured_as_written_ms_per_step': None if ref_loop is None else ref_loop.get('captured_as_written_ms_per_step'),
            'reference_loop': ref_loop,
            'parity': _parity_report(),
        }
        out['roofline'] = roofline_meanfield(data, dev, 8, T)
        out['roofline']['measured_copy_GBps'] = copy_ceiling(dev)
        out['roofline_bwd'] = roofline_meanfield_bwd(data, dev, 8, T)
        if out['roofline_bwd'] is not None:
            # forward and backward of the level-0 layer as ONE figure: algorithmic bytes of both over the sum of their
            # average times (each timed on its own, back to back on the launching stream)
            rf, rb = out['roofline'], out['roofline_bwd']
            alg = rf['alg_bytes_per_launch'] + rb['alg_bytes_per_launch']
            t = (rf['avg_launch_us'] + rb['avg_launch_us']) * 1e-6
            tr = None if rf['traffic'] is None or rb['traffic'] is None else rf['traffic'] + rb['traffic']
            out['roofline_fwd_bwd'] = {'bound': 'hbm', 'achieved': alg / t / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                                       'frac': alg / t / HBM_PEAK, 'traffic': tr, 'alg_bytes_per_launch': alg,
                                       'avg_launch_us': t * 1e6,
                                       'kernel': 'level-0 mean-field forward + backward (roofline + roofline_bwd)'}
        try:
            out['roofline_layer'] = roofline_layer(data, dev, T, level0=(out['roofline'], out['roofline_bwd']) if out['roofline_bwd'] else None)
        except Exception as e:                             # a measurement beside the contract line, never fatal
            out['roofline_layer'] = {'error': str(e).splitlines()[0][:200]}
        try:
            out['roofline_pointconv'] = roofline_pointconv(data, dev, 8)
        except Exception as e:                             # a measurement beside the contract line, never fatal
            out['roofline_pointconv'] = {'error': str(e).splitlines()[0][:200]}
        # the whole step against the HBM roofline: algorithmic bytes (step_byte_model + 20 B per parameter for SGD with momentum) beside
        # the HBM-side traffic measured by rocprofv3 PMC passes of this build (profiles/, null when the sources changed since)
        n_par = sum(p.numel() for p in net.parameters())
        alg, groups = step_byte_model(B, N, 16, T, n_cls)
        alg += 20 * n_par
        step_traffic, step_note = _measured_traffic(TRAFFIC_STEP, {'B': B, 'N': N, 'K': 16, 'T': T})
        out['roofline_step'] = {'bound': 'hbm', 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'alg_bytes_per_step': alg,
                                'achieved': alg / (ms_per_step * 1e-3) / 1e9, 'frac': alg / (ms_per_step * 1e-3) / HBM_PEAK,
                                'traffic': step_traffic, 'traffic_source': step_note,
                                'waste_ratio': None if step_traffic is None else step_traffic / alg,
                                'frac_on_measured_traffic': None if step_traffic is None else step_traffic / (ms_per_step * 1e-3) / HBM_PEAK,
                                'groups': groups, 'ms_per_step': ms_per_step,
                                'kernel': 'the whole captured training step (fwd + weighted CE + bwd + SGD), %d parameters' % n_par}
        if world == 1 and not args.no_other_configs:          # every BASELINE config in the line: a few seconds on one GPU
            try:
                out['other_configs'] = other_configs(dev)
            except Exception as e:
                import traceback
                traceback.print_exc()
                torch.cuda.synchronize()
                out['other_configs'] = {'error': str(e).splitlines()[0][:200]}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(data, net, T, data.y, n_cls, dev)
        print(json.dumps(out))
    if grouped:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
