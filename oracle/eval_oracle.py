"""CPU restatements (numpy) of the callers either side of the network -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product
(crfconv_amd/) never does.  Each function cites the reference lines it follows.

Pinned: ``fast_hist`` / ``scores`` / ``shapenet_part_iou`` (g11) against utils/metrics.py run from /root/reference, and ``possibility_draw``
against ``Semantic3D._get_random`` run from /root/reference with sklearn's KDTree (tests/golden/g9_eval.npz, made by
tests/golden/make_golden.py).  ``iou_from_confusions`` and ``vote_update`` / ``vote_project`` restate
trainval.py:76-90 and :186-203, whose module cannot be imported here (it pulls torch_points3d / torch_geometric
transforms at module scope): parity unpinned for those three, they are a few numpy expressions each.
"""
import numpy as np


def fast_hist(label_true, label_pred, n_class, ignore_index=-1):
    """utils/metrics.py:13-19."""
    label_true = np.asarray(label_true).reshape(-1)
    label_pred = np.asarray(label_pred).reshape(-1)
    keep = (label_true >= 0) & (label_true < n_class) & (label_true != ignore_index)
    flat = n_class * label_true[keep].astype(np.int64) + label_pred[keep]
    return np.bincount(flat, minlength=n_class * n_class).reshape(n_class, n_class)


def scores(hist):
    """utils/metrics.py:28-56 on a confusion matrix (rows = truth)."""
    hist = np.asarray(hist, dtype=np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        d = np.diag(hist)
        acc = d.sum() / hist.sum()
        acc_cls = np.nanmean(d / hist.sum(axis=1))
        iu = d / (hist.sum(axis=1) + hist.sum(axis=0) - d)
        freq = hist.sum(axis=1) / hist.sum()
        fw = (freq[freq > 0] * iu[freq > 0]).sum()
    return {'Overall Acc': acc, 'Mean Acc': acc_cls, 'FreqW Acc': fw, 'Mean IoU': np.nanmean(iu)}, iu


def iou_from_confusions(confusions):
    """trainval.py:76-90."""
    c = np.asarray(confusions, dtype=np.float64)
    tp = np.diagonal(c, axis1=-2, axis2=-1)
    tpfn = c.sum(-1)
    tpfp = c.sum(-2)
    iou = tp / (tpfp + tpfn - tp + 1e-6)
    mask = tpfn < 1e-3
    counts = np.sum(1 - mask, axis=-1, keepdims=True)
    miou = np.sum(iou, axis=-1, keepdims=True) / (counts + 1e-6)
    return iou + mask * miou


def vote_update(test_probs, point_idx, prob, smooth):
    """trainval.py:186-189 for one sample: float32 table, python-float coefficients (weak scalars -> float32)."""
    test_probs[point_idx] = smooth * test_probs[point_idx] + (1 - smooth) * prob.astype(np.float32)
    return test_probs


def vote_project(test_probs, proj_idx, label_offset=1):
    """trainval.py:200-203."""
    return np.argmax(test_probs[proj_idx, :], axis=1).astype(np.uint8) + label_offset


def softmax32(logits):
    z = logits.astype(np.float32)
    e = np.exp(z - z.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


def possibility_draw(points, possibility, num_points, noise, weights=None):
    """datasets/semantic3d_dataset.py:424-451 for one cloud (cloud choice = arg-min of the per-cloud minima is the
    caller's).  points float32 [n, 3] (the KD-tree holds them as float64); possibility float64 [n], updated in
    place; noise float64 [3] = the Gaussian jitter; weights float64 [n] per point or None (test split).
    Returns (query_idx sorted by (distance, index), centred xyz float32 in that order, pick_point float64)."""
    pts = points.astype(np.float64)
    pick_idx = int(np.argmin(possibility))
    pick = pts[pick_idx].reshape(1, -1) + noise.reshape(1, -1)
    d64 = np.sum(np.square(pts - pick), axis=1)
    order = np.lexsort((np.arange(len(pts)), d64))[:num_points]          # k nearest, ties by index
    xyz = pts[order].copy()
    xyz[:, 0:2] = xyz[:, 0:2] - pick[:, 0:2]
    dists = np.sum(np.square(pts[order] - pick).astype(np.float32), axis=1)
    delta = np.square(1 - dists / np.max(dists)) * (1 if weights is None else weights[order])
    possibility[order] += delta
    return order, xyz.astype(np.float32), pick.reshape(-1)


def shapenet_part_iou(label_trues, label_preds, parts):
    """utils/metrics.py:87-103 (runningScoreShapeNet.update) for one shape: mean over the category's part labels of
    (|true & pred| + eps) / (|true | pred| + eps), eps = float32 machine epsilon."""
    eps = np.finfo(np.float32).eps
    total = 0.0
    for l in parts:
        t, p = (label_trues == l), (label_preds == l)
        total += (np.sum(np.logical_and(t, p)) + eps) / (np.sum(np.logical_or(t, p)) + eps)
    return total / len(parts)
