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
from .graph import ptr, require_gpu, stream_ptr


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
        c = int(torch.argmin(self._minv).item())           # :424 -- the loop's only host decision (which cloud)
        pts = self.points[c]
        n, k = pts.shape[0], min(self.num_points, pts.shape[0])
        if noise is None:
            noise = torch.randn(3, dtype=torch.float64, generator=self.generator) * self.noise_scale
        noise = torch.as_tensor(noise, dtype=torch.float64).to(self.device).contiguous()
        if perm is None:
            perm = torch.randperm(k, generator=self.generator)
        perm = None if perm is False else torch.as_tensor(perm, dtype=torch.int64).to(self.device).contiguous()
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
        return out


class VoteAccumulator:
    """``test_probs`` of trainval.py:58 (one float32 [n_c, n_classes] table per cloud, zeros) with the update of
    :186-189 and the projection of :198-203."""

    def __init__(self, cloud_sizes, num_classes, smooth=0.98, device='cuda'):
        self.num_classes = int(num_classes)
        self.smooth = float(smooth)
        self.test_probs = [torch.zeros((int(n), self.num_classes), dtype=torch.float32, device=device) for n in cloud_sizes]
        self._bad = torch.zeros(1, dtype=torch.int32, device=device)

    def update(self, point_idx, cloud_idx, probs=None, logits=None):
        """point_idx int64 [B, N]; cloud_idx int64 [B] or [B, 1]; probs or logits float32 [B * N, C] (the network's
        output layout).  Samples are applied in batch order, as the reference's ``for b in range(batch_size)``."""
        src = probs if probs is not None else logits
        require_gpu(point_idx, src)
        B, N = point_idx.shape
        src = src.reshape(B, N, self.num_classes).float().contiguous()
        point_idx = point_idx.long().contiguous()
        clouds = cloud_idx.reshape(B, -1)[:, 0].tolist()
        for b in range(B):
            tp = self.test_probs[int(clouds[b])]
            _lib.call('crfconv_vote_accumulate', ptr(src[b]) if probs is not None else None,
                      ptr(src[b]) if probs is None else None, ptr(point_idx[b]), N, self.num_classes, self.smooth,
                      ptr(tp), tp.shape[0], ptr(self._bad), stream_ptr())

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
