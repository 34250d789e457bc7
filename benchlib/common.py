"""Shared pieces of the benchmark: the synthetic config-2 clouds, timing helpers, the committed profiler records bench.py quotes."""
import json
import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8.0e12           # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
TRAFFIC_FWD, TRAFFIC_BWD = 'r6_meanfield_traffic.json', 'r6_meanfield_bwd_traffic.json'      # PMC passes, sha1-keyed to the kernel sources
TRAFFIC_STEP, TRAFFIC_PC = 'r6_step_traffic.json', 'r6_pointconv_traffic.json'
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


