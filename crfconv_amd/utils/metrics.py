"""``runningScore`` of the reference (utils/metrics.py:7-56) with the confusion matrix kept on the GPU.

Same constructor, ``update`` / ``get_scores`` / ``reset`` and score names; ``update`` takes the CUDA tensors the
training loop already holds (trainval.py:108 moves both to the host every step -- a device sync per step) and
``update_from_logits`` fuses the arg-max.  Only ``get_scores`` / ``confusion_matrix`` copy the n x n counts back.
"""
import numpy as np
import torch

from .. import _lib
from ..graph import ptr, require_gpu, stream_ptr


class runningScore(object):
    def __init__(self, n_classes, ignore_index=-1, device='cuda'):
        self.n_classes = int(n_classes)
        self.ignore_index = int(ignore_index)
        self._hist = torch.zeros((self.n_classes, self.n_classes), dtype=torch.int64, device=device)
        self._bad = torch.zeros(1, dtype=torch.int32, device=device)

    def _accumulate(self, label_trues, label_preds, logits, label_shift):
        require_gpu(label_trues, label_preds, logits)
        lt = label_trues.reshape(-1).contiguous()
        if lt.dtype != torch.int64:
            lt = lt.long()
        if logits is not None:
            if logits.shape[-1] != self.n_classes:
                raise ValueError('logits have %d classes, expected %d' % (logits.shape[-1], self.n_classes))
            logits = logits.reshape(-1, self.n_classes).float().contiguous()
            n = logits.shape[0]
        else:
            label_preds = label_preds.reshape(-1).contiguous()
            if label_preds.dtype != torch.int64:
                label_preds = label_preds.long()
            n = label_preds.numel()
        if n != lt.numel():
            raise ValueError('%d labels for %d predictions' % (lt.numel(), n))
        _lib.call('crfconv_confusion_accumulate', ptr(lt), ptr(label_preds), ptr(logits), n, self.n_classes,
                  self.ignore_index, int(label_shift), ptr(self._hist), ptr(self._bad), stream_ptr())

    def update(self, label_trues, label_preds):
        """metrics.py:21-26: rows of a 2-D input are accumulated one by one there, which equals the flat histogram."""
        self._accumulate(label_trues, label_preds, None, 0)

    def update_from_logits(self, label_trues, logits, label_shift=0):
        """``update(y, logits.max(dim=1)[1])`` (trainval.py:108) in one pass; ``label_shift=1`` folds the loop's
        ``data.y.reshape(-1) - 1`` (:100)."""
        self._accumulate(label_trues, None, logits, label_shift)

    @property
    def confusion_matrix(self):
        if int(self._bad.item()):
            raise ValueError('%d predictions outside [0, %d)' % (int(self._bad.item()), self.n_classes))
        return self._hist.cpu().numpy().astype(np.float64)

    def get_scores(self):
        """The four scores and the per-class IoU table of metrics.py:28-56, from the host copy of the counts (one
        sync).  Rows = ground truth, columns = prediction; classes with an empty row / union give nan and drop out
        of the nan-means, exactly as there."""
        cm = self.confusion_matrix
        hit = np.diag(cm)
        n_true, n_pred, total = cm.sum(axis=1), cm.sum(axis=0), cm.sum()
        with np.errstate(divide='ignore', invalid='ignore'):
            iou = hit / (n_true + n_pred - hit)
            share = n_true / total
            scores = {
                'Overall Acc': hit.sum() / total,
                'Mean Acc': np.nanmean(hit / n_true),
                'FreqW Acc': (share[share > 0] * iou[share > 0]).sum(),
                'Mean IoU': np.nanmean(iou),
            }
        return scores, {c: iou[c] for c in range(self.n_classes)}

    def reset(self):
        self._hist.zero_()
        self._bad.zero_()


class runningScoreShapeNet(object):
    """Per-shape part IoU of the reference (utils/metrics.py:59-119): ``update(label_trues, label_preds, category)``
    scores ONE shape -- the mean over the parts of its object category of |true & pred| / |true | pred|, both counts
    offset by float32 eps as there -- and ``get_scores()`` returns (instance mIoU, category mIoU, per-category table).
    The per-part intersections / unions come from the device confusion kernel (one 50 x 50 histogram per shape)."""

    obj_classes = {'Airplane': 0, 'Bag': 1, 'Cap': 2, 'Car': 3, 'Chair': 4, 'Earphone': 5, 'Guitar': 6, 'Knife': 7,
                   'Lamp': 8, 'Laptop': 9, 'Motorbike': 10, 'Mug': 11, 'Pistol': 12, 'Rocket': 13, 'Skateboard': 14,
                   'Table': 15}
    seg_classes = {'Earphone': [16, 17, 18], 'Motorbike': [30, 31, 32, 33, 34, 35], 'Rocket': [41, 42, 43],
                   'Car': [8, 9, 10, 11], 'Laptop': [28, 29], 'Cap': [6, 7], 'Skateboard': [44, 45, 46],
                   'Mug': [36, 37], 'Guitar': [19, 20, 21], 'Bag': [4, 5], 'Lamp': [24, 25, 26, 27],
                   'Table': [47, 48, 49], 'Airplane': [0, 1, 2, 3], 'Pistol': [38, 39, 40],
                   'Chair': [12, 13, 14, 15], 'Knife': [22, 23]}
    N_PARTS = 50

    def __init__(self, device='cuda'):
        self._names = {v: k for k, v in self.obj_classes.items()}
        self._score = runningScore(self.N_PARTS, ignore_index=-1, device=device)
        self.category_IoU = np.zeros(16, dtype=np.float32)
        self.category_num = np.zeros(16, dtype=np.int32)

    def update(self, label_trues, label_preds, category):
        parts = self.seg_classes[self._names[int(category)]]
        self._score.reset()
        self._score.update(label_trues, label_preds)
        cm = self._score.confusion_matrix
        eps = np.finfo(np.float32).eps
        iu = 0.0
        for l in parts:
            inter = cm[l, l]
            union = cm[l, :].sum() + cm[:, l].sum() - inter
            iu += (inter + eps) / (union + eps)
        iu /= len(parts)
        self.category_IoU[int(category)] += iu
        self.category_num[int(category)] += 1
        return iu

    def get_scores(self):
        with np.errstate(divide='ignore', invalid='ignore'):
            per_class = self.category_IoU / self.category_num
            pIoU = self.category_IoU.sum() / self.category_num.sum()
        return pIoU, per_class.mean(), {k: per_class[v] for k, v in self.obj_classes.items()}


def iou_from_confusions(confusions, eps=1e-6):
    """Trainer._iou_from_confusions (trainval.py:76-90): per-class IoU from [..., n, n] confusion counts, where a
    class that never occurs in the ground truth (row sum < 1e-3) is given the mean IoU of the classes that do."""
    cm = np.asarray(confusions, dtype=np.float64)
    hit = np.diagonal(cm, axis1=-2, axis2=-1)
    n_true, n_pred = cm.sum(axis=-1), cm.sum(axis=-2)
    iou = hit / (n_pred + n_true - hit + eps)
    absent = n_true < 1e-3
    present = (~absent).sum(axis=-1, keepdims=True)
    mean_present = iou.sum(axis=-1, keepdims=True) / (present + eps)
    return iou + absent * mean_present
