"""The CPU baseline leg (oracle / reference cores timed on this host) and the committed parity report."""
import json
import os

import numpy as np
import torch

from .common import ROOT, _median_time


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


