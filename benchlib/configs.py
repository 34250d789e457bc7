"""The other BASELINE.json configs on one GPU, and configs 5 / 3 as pipelines."""
import time

import numpy as np
import torch

from .common import _event_time, synth_cloud
from .roofline import _layer_summary, roofline_layer


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

    cache = {}

    def run(timings, graphed=False, keep=False):
        smp = PossibilitySampler([pts], rgb=[rgb], num_points=n_crop, split='test', generator=torch.Generator().manual_seed(51))
        votes = VoteAccumulator([n_scene], C, device=dev)
        vote_scene(smp, net, votes, n_crops, kernel_size=(K,) * 5, generator=torch.Generator().manual_seed(52), timings=timings, graphed=graphed,
                   graph_cache=cache if keep else None)
        return votes
    run(None)                                              # warm-up (allocator, lazily built tables, kernel modules)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    votes = run(None)
    torch.cuda.synchronize()
    t_loop = time.perf_counter() - t0
    stages = {}
    run(stages)                                            # the same again with a device synchronisation around every stage
    run(None, graphed=True)                                # collate + forward of crops 2 .. 16 as hipGraph replays (the first crop captures)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    votes_g = run(None, graphed=True)
    torch.cuda.synchronize()
    t_loop_g = time.perf_counter() - t0
    stages_g = {}
    run(stages_g, graphed=True)
    del votes_g
    run(None, graphed=True, keep=True)                     # ... and with the graphs kept across scenes (graph_cache): no capture in the timed run
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(None, graphed=True, keep=True)
    torch.cuda.synchronize()
    t_loop_gk = time.perf_counter() - t0
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
        'graphed_loop_ms': t_loop_g * 1e3, 'graphed_scene_points_per_s_M': n_scene / (t_loop_g + t_project) / 1e6,
        'graphed_stage_ms_total_synchronised': stages_g,
        'graphed_cached_loop_ms': t_loop_gk * 1e3, 'graphed_cached_scene_points_per_s_M': n_scene / (t_loop_gk + t_project) / 1e6,
        'stage_ms_per_crop_synchronised': {k: v / n_crops for k, v in stages.items()},
        'project_ms': t_project * 1e3, 'raw_points_projected': int(raw.shape[0]), 'projection_index_ms_offline': t_proj_idx * 1e3,
        'scene_fraction_voted': covered, 'labels_histogram': torch.bincount(labels.long(), minlength=C + 1).tolist(),
        'what': 'sampler -> multiscale_compute(K=32) -> PointConvBig(T=5).eval() -> votes for 16 crops (eager, B = 1 per crop as the sampler yields '
                'them), then the arg-max re-projection onto a raw cloud; graphed_*: vote_scene(graphed=True) -- the first crop runs eagerly and captures, '
                'crops 2 .. 16 are two replays each, the capture included in graphed_loop_ms (graphed_cached_*: graphs kept across scenes, graph_cache=, no capture); loop_ms is wall time without per-stage synchronisation, the stage '
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


