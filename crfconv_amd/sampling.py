"""The two callers either side of tiled-scene inference, on the device:

* ``PossibilitySampler`` -- the crop sampler of datasets/semantic3d_dataset.py:423-460 (``Semantic3D._get_random``;
  the S3DIS copy is s3dis_dataset.py:338-395): seed = arg-min possibility over clouds and points, Gaussian jitter,
  crop = the ``num_points`` nearest points, possibility += (1 - d / d_max)^2 * class weight.
* ``VoteAccumulator`` -- the running mean of soft-max votes per cloud point and the re-projection arg-max of
  trainval.py:170-214.

The reference does both on the host with sklearn's KDTree and numpy; here the clouds, possibilities and vote tables
stay in HBM and each call enqueues a handful of kernels of libcrfconv_amd.so (csrc/evaluate.hip).
"""
import numpy as np
import torch

from . import _lib
from .data import Data
from .graph import ptr, require_gpu, stream_ptr, to_device


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


class PossibilitySampler:
    """points: list of float32 [n_c, 3] CUDA tensors (the sub-sampled clouds); rgb / labels: matching lists or None.

    ``class_weight`` float64 [n_classes] and ``label_to_idx`` (dict raw label -> class id) give the per-point update
    weight of the train / val splits (:443-445); ``split='test'`` uses weight 1 and zero labels (:439-441).
    Possibilities start as ``randn * 1e-3`` (:267) from ``generator``, or from ``possibility`` if given."""

    def __init__(self, points, rgb=None, labels=None, num_points=65536, class_weight=None, label_to_idx=None,
                 split='train', generator=None, possibility=None, noise_scale=3.5 / 10):
        require_gpu(*points)
        self.points = [p.float().contiguous() for p in points]
        self.device = self.points[0].device
        self.rgb = rgb
        self.split = split
        self.num_points = int(num_points)
        self.noise_scale = float(noise_scale)
        self.generator = generator
        self.labels = None
        self.point_weight = None
        if split != 'test' and labels is not None:
            self.labels, self.point_weight = [], []
            cw = torch.as_tensor(np.asarray(class_weight, dtype=np.float64)).to(self.device)
            for lab in labels:
                lab = lab.to(self.device).long()
                if label_to_idx is not None:
                    raw = torch.tensor(sorted(label_to_idx), device=self.device)
                    cls = torch.tensor([label_to_idx[k] for k in sorted(label_to_idx)], device=self.device)
                    lab = cls[torch.searchsorted(raw, lab)]
                self.labels.append(lab)
                self.point_weight.append(cw[lab].contiguous())
        if possibility is not None:
            self.possibility = [torch.as_tensor(p, dtype=torch.float64).to(self.device).contiguous().clone()
                                for p in possibility]
        else:
            self.possibility = [(torch.randn(p.shape[0], dtype=torch.float64, generator=generator) * 1e-3).to(self.device)
                                for p in self.points]
        self._ws_min = _ws(_lib.load().crfconv_argmin_workspace(), self.device)
        self._minv = torch.empty(len(self.points), dtype=torch.float64, device=self.device)
        self._mini = torch.empty(len(self.points), dtype=torch.int64, device=self.device)
        for c in range(len(self.points)):
            self._refresh_min(c)
        self._crop_ws = {}

    def _refresh_min(self, c):
        p = self.possibility[c]
        _lib.call('crfconv_argmin_f64', ptr(p), p.numel(), ptr(self._minv[c:]), ptr(self._mini[c:]), ptr(self._ws_min),
                  self._ws_min.numel(), stream_ptr())

    @property
    def min_possibility(self):
        return self._minv.cpu().numpy()

    def get_random(self, noise=None, perm=None):
        """One crop.  ``noise`` float64 [3] and ``perm`` int64 [k] override the Gaussian jitter and the shuffle (tests
        feed the reference's draws); otherwise they come from ``self.generator``.  Returns ``Data(pos, rgb, y,
        point_idx, cloud_idx)`` with the reference's field meanings (:453-458), all on the device."""
        # :424 -- which cloud: the loop's only host decision (a device -> host read; none with a single cloud)
        c = 0 if len(self.points) == 1 else int(torch.argmin(self._minv).item())
        pts = self.points[c]
        n, k = pts.shape[0], min(self.num_points, pts.shape[0])
        if noise is None:
            noise = torch.randn(3, dtype=torch.float64, generator=self.generator) * self.noise_scale
        noise = to_device(torch.as_tensor(noise, dtype=torch.float64).contiguous(), self.device)
        if perm is None:
            perm = torch.randperm(k, generator=self.generator)
        perm = None if perm is False else to_device(torch.as_tensor(perm, dtype=torch.int64).contiguous(), self.device)
        key = (n, k)
        if key not in self._crop_ws:
            self._crop_ws[key] = _ws(_lib.load().crfconv_possibility_crop_workspace(n, k), self.device)
        ws = self._crop_ws[key]
        idx = torch.empty(k, dtype=torch.int64, device=self.device)
        xyz = torch.empty((k, 3), dtype=torch.float32, device=self.device)
        center = torch.empty(3, dtype=torch.float64, device=self.device)
        pw = None if self.point_weight is None else self.point_weight[c]
        _lib.call('crfconv_possibility_crop', ptr(pts), n, k, ptr(self._mini[c:]), ptr(noise), ptr(perm), ptr(pw),
                  ptr(self.possibility[c]), ptr(idx), ptr(xyz), ptr(center), ptr(ws), ws.numel(), stream_ptr())
        self._refresh_min(c)                               # :451
        rgb = None if self.rgb is None else self.rgb[c][idx].float()
        if self.labels is None:
            y = torch.zeros(k, dtype=torch.long, device=self.device)
        else:
            y = self.labels[c][idx]
        out = Data(pos=xyz, rgb=rgb, y=y, point_idx=idx, cloud_idx=torch.tensor([c], dtype=torch.long, device=self.device))
        out.center = center
        out.cloud = c                                        # the same as a host int (VoteAccumulator.update takes it without a device read)
        return out


class VoteAccumulator:
    """``test_probs`` of trainval.py:58 (one float32 [n_c, n_classes] table per cloud, zeros) with the update of
    :186-189 and the projection of :198-203."""

    def __init__(self, cloud_sizes, num_classes, smooth=0.98, device='cuda', track_visits=False):
        """track_visits: count the updates of every point (int32 tables beside the votes) -- what ``merge`` needs when the crops of a
        scene are spread over several accumulators (ranks)."""
        self.num_classes = int(num_classes)
        self.smooth = float(smooth)
        self.test_probs = [torch.zeros((int(n), self.num_classes), dtype=torch.float32, device=device) for n in cloud_sizes]
        self.visits = [torch.zeros(int(n), dtype=torch.int32, device=device) for n in cloud_sizes] if track_visits else None
        self._bad = torch.zeros(1, dtype=torch.int32, device=device)

    def update(self, point_idx, cloud_idx, probs=None, logits=None):
        """point_idx int64 [B, N]; cloud_idx int64 [B] or [B, 1]; probs or logits float32 [B * N, C] (the network's
        output layout).  Samples are applied in batch order, as the reference's ``for b in range(batch_size)``."""
        src = probs if probs is not None else logits
        require_gpu(point_idx, src)
        B, N = point_idx.shape
        src = src.reshape(B, N, self.num_classes).float().contiguous()
        point_idx = point_idx.long().contiguous()
        # cloud ids: host ints / a list of them are taken as they are; a device tensor costs one device -> host read per call
        clouds = cloud_idx.reshape(B, -1)[:, 0].tolist() if torch.is_tensor(cloud_idx) else ([int(cloud_idx)] if isinstance(cloud_idx, int) else [int(c) for c in cloud_idx])
        for b in range(B):
            tp = self.test_probs[int(clouds[b])]
            if self.visits is None:
                _lib.call('crfconv_vote_accumulate', ptr(src[b]) if probs is not None else None,
                          ptr(src[b]) if probs is None else None, ptr(point_idx[b]), N, self.num_classes, self.smooth,
                          ptr(tp), tp.shape[0], ptr(self._bad), stream_ptr())
            else:
                _lib.call('crfconv_vote_accumulate_counted', ptr(src[b]) if probs is not None else None,
                          ptr(src[b]) if probs is None else None, ptr(point_idx[b]), N, self.num_classes, self.smooth,
                          ptr(tp), tp.shape[0], ptr(self._bad), ptr(self.visits[int(clouds[b])]), stream_ptr())

    def fold_(self, later_probs, later_visits):
        """self <- the tables ONE accumulator would hold that applied self's updates first and then those behind `later_probs` /
        `later_visits` (lists like self.test_probs / self.visits): trainval.py:188-189 is a running mean v <- s v + (1 - s) p, and
        n later updates of a point scale what was there by s^n (csrc/evaluate.hip: vote_fold_kernel)."""
        if self.visits is None:
            raise _lib.CrfConvError('VoteAccumulator.fold_ needs track_visits=True on both sides')
        for tp, vi, lp, lv in zip(self.test_probs, self.visits, later_probs, later_visits):
            lp = lp.to(tp.device, torch.float32).contiguous()
            lv = lv.to(tp.device, torch.int32).contiguous()
            if lp.shape != tp.shape or lv.shape != vi.shape:
                raise ValueError('fold_: tables of %s / %s against %s / %s' % (tuple(lp.shape), tuple(lv.shape), tuple(tp.shape), tuple(vi.shape)))
            _lib.call('crfconv_vote_fold', ptr(tp), ptr(vi), ptr(lp), ptr(lv), tp.shape[0], self.num_classes, self.smooth, stream_ptr())

    def merge(self, group=None):
        """Crops of one scene sharded over the ranks of a process group (each rank voted its own crops into its own tables): every rank
        leaves with the SAME merged tables -- those of a single accumulator that saw rank 0's crops first, then rank 1's, ... (the
        reference's update is order-dependent; sharding only fixes this order, it does not change the rule).  One all-gather of the
        tables and the visit counts per cloud, folded in rank order.  No-op without a process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return self
        if self.visits is None:
            raise _lib.CrfConvError('VoteAccumulator.merge needs track_visits=True')
        world = dist.get_world_size(group)
        on_host = dist.get_backend(group) == 'gloo'            # (gloo moves host tensors; RCCL device tensors)
        for c, (tp, vi) in enumerate(zip(self.test_probs, self.visits)):
            send_p, send_v = (tp.cpu(), vi.cpu()) if on_host else (tp, vi)
            all_p = [torch.empty_like(send_p) for _ in range(world)]
            all_v = [torch.empty_like(send_v) for _ in range(world)]
            dist.all_gather(all_p, send_p, group=group)
            dist.all_gather(all_v, send_v, group=group)
            tp.copy_(all_p[0])
            vi.copy_(all_v[0])
            for r in range(1, world):
                _lib.call('crfconv_vote_fold', ptr(tp), ptr(vi), ptr(all_p[r].to(tp.device).contiguous()),
                          ptr(all_v[r].to(tp.device).contiguous()), tp.shape[0], self.num_classes, self.smooth, stream_ptr())
        return self

    def project(self, cloud, proj_idx, label_offset=1):
        """uint8 labels of the original points: arg-max of the votes of their nearest sub-sampled point, + 1 because
        0 means unlabeled (:201-203)."""
        require_gpu(proj_idx)
        tp = self.test_probs[cloud]
        proj_idx = proj_idx.long().contiguous()
        preds = torch.empty(proj_idx.numel(), dtype=torch.uint8, device=tp.device)
        _lib.call('crfconv_vote_project', ptr(tp), ptr(proj_idx), proj_idx.numel(), self.num_classes, tp.shape[0],
                  int(label_offset), ptr(preds), ptr(self._bad), stream_ptr())
        return preds

    def check(self):
        bad = int(self._bad.item())
        if bad:
            raise IndexError('%d point indices outside their cloud' % bad)


class _GraphedCrops:
    """Collate + eval forward of one crop SHAPE as two hipGraph replays (vote_scene(graphed=True)): the first crop of a shape runs eagerly and
    becomes the static batch; ``data.CollateGraph`` (Morton order, kNN at every scale, counter-based subsets, in-place refresh of the static
    batch's tables) and the captured ``net(static)`` serve every later crop of that shape.  Eagerly a crop is ~250 library launches of
    host-bound Python (4.7 ms of network + 2.5 ms of collate at 65 536 points, K = 32, T = 5); replayed it is their kernel time."""

    def __init__(self, net, first, kernel_size, ratio, generator):
        from .data import CollateGraph
        first.point_idx = first.cloud_idx = None         # (per-crop bookkeeping, not inputs of the network: the caller keeps them)
        self.static = first
        self.cg = CollateGraph(first, kernel_size=kernel_size, ratio=ratio, generator=generator)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            net(first)                                   # warm-up outside the capture (lazily built tables, allocator)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.logits = net(first)

    def run(self, pos, x):
        self.cg.run(pos, x)                              # static batch <- this crop (its own Morton order: self.cg.order)
        self.graph.replay()
        return self.logits, self.cg.order


def vote_scene(sampler, net, votes, n_crops, kernel_size=(16, 16, 16, 16, 16), ratio=(4, 4, 4, 4, 2), rank=0, world=1, generator=None,
               timings=None, graphed=False, on_crop=None, graph_cache=None):
    """The inference loop of trainval.py:170-189 on the device for ``n_crops`` crops of the sampler's clouds: crop (possibility
    sampler) -> ``multiscale_compute`` (kNN at every scale) -> ``net`` (eval, no grad) -> soft-max votes into ``votes``.  One crop per
    batch (B = 1: datasets/semantic3d_dataset.py:453-458 yields single crops; the loader's batch dimension only stacks them).
    Features = [xyz, rgb] as the reference's collate builds them (datasets/semantic3d_dataset.py:507-510).

    world > 1: EVERY rank draws the whole crop sequence (the sampler is cheap and its possibilities must evolve as on one GPU), crop i is
    run through the network by rank i % world only; ``votes.merge()`` afterwards gives every rank the full tables.
    graphed: collate and forward of every crop after the first of its shape as hipGraph replays (_GraphedCrops); the random subsets of the
    coarse levels then come from the collate graph's counter-based draw instead of ``torch.randperm`` (any subset is a valid one).
    graph_cache: a dict the captured graphs are kept in across calls (same net, same crop shapes: further scenes pay no capture).
    timings: a dict that receives the summed milliseconds per stage (host clock around device-synchronised stages: a diagnostic mode --
    it serialises host and device).  on_crop(data, logits, point_idx): called per crop with the collated batch (static buffers when graphed:
    clone what is to be kept), the logits and the crop's point ids in the batch's row order (tests)."""
    import time
    from .data import multiscale_compute
    dev = sampler.device

    def stage(name, fn):
        if timings is None:
            return fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        timings[name] = timings.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return out
    was_training = net.training
    net.eval()
    graphs = graph_cache if graph_cache is not None else {}
    try:
        for i in range(n_crops):
            crop = stage('sample', sampler.get_random)
            if i % world != rank:
                continue
            pos = crop.pos.unsqueeze(0)
            rgb = crop.rgb if crop.rgb is not None else torch.zeros_like(crop.pos)
            x = torch.cat([crop.pos, rgb], -1).unsqueeze(0)
            shape = tuple(pos.shape)
            if graphed and shape in graphs:
                logits, order = stage('collate+network (replays)', lambda: graphs[shape].run(pos, x))
                data = graphs[shape].static
                point_idx = crop.point_idx[order.reshape(-1)].unsqueeze(0)          # the batch's rows are the crop's points in Morton order
            else:
                data = stage('collate', lambda: multiscale_compute(pos, x=x, point_idx=crop.point_idx.unsqueeze(0), cloud_idx=crop.cloud_idx.reshape(1, 1),
                                                                   kernel_size=kernel_size, ratio=ratio, generator=generator, sort='morton'))
                with torch.no_grad():
                    logits = stage('network', lambda: net(data))
                # the collate reordered the crop along its Morton curve: point_idx travelled with it (multiscale_compute permutes x, y, point_idx alike)
                point_idx = data.point_idx
                if graphed:
                    votes.update(point_idx, crop.cloud, logits=logits)
                    if on_crop is not None:
                        on_crop(data, logits, point_idx)
                    graphs[shape] = _GraphedCrops(net, data, kernel_size, ratio, generator)
                    continue
            stage('vote', lambda: votes.update(point_idx, crop.cloud, logits=logits))
            if on_crop is not None:
                on_crop(data, logits, point_idx)
    finally:
        net.train(was_training)
    return votes
