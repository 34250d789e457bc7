"""Device-side graph builders with the call signatures of the torch_cluster / torch_geometric functions the
reference's sparse path uses (models/point_conv.py:5, 341-396; models/continuous_crf_conv.py:53):
knn_graph, knn, radius_graph, fps -- all on top of the exact HIP grid kNN (csrc/knn.hip).

Conventions (PyG): an edge_index is [2, E] with row 0 = source j ("col"), row 1 = target i ("row").
Third-party semantics are unpinned (no version, no tests in the reference): knn_* are exact; radius_graph keeps
the NEAREST `max_num_neighbors` inside the radius (torch_cluster keeps the first it meets); fps starts from the
first point of each cloud unless random_start=True."""
import torch

from ..utils import nearest_neighbors as nn_


def _segments(batch, n):
    """[(start, end)] of each cloud; `batch` must be sorted (PyG convention). None -> one cloud."""
    if batch is None:
        return [(0, n)]
    counts = torch.bincount(batch).tolist()
    out, o = [], 0
    for c in counts:
        if c:
            out.append((o, o + c))
        o += c
    return out


def _knn_segments(x, y, k, seg_x, seg_y):
    """For every y row the k nearest x rows of the same cloud -> (y_index, x_index) global ids, nearest first."""
    rows, cols = [], []
    for (xs, xe), (ys, ye) in zip(seg_x, seg_y):
        kk = min(k, xe - xs)
        idx = nn_.knn_batch_device(x[xs:xe].unsqueeze(0), y[ys:ye].unsqueeze(0), kk)[0]     # [ny, kk]
        rows.append((torch.arange(ys, ye, device=x.device).unsqueeze(1).expand(-1, kk)).reshape(-1))
        cols.append((idx + xs).reshape(-1))
    return torch.cat(rows), torch.cat(cols)


def knn_dilated(x, y, k, dilation, batch_x=None, batch_y=None, generator=None):
    """The reference's random dilation (models/point_conv.py:357-364, 387-394): search k * dilation neighbours, keep k
    of them per query, drawn with replacement by torch.randint.  Done per cloud so that a cloud with fewer than
    k * dilation points (the coarse levels of small clouds) draws from the neighbours it has instead of indexing
    past them.  -> (y_index, x_index)."""
    rows, cols = [], []
    for (xs, xe), (ys, ye) in zip(_segments(batch_x, x.shape[0]), _segments(batch_y, y.shape[0])):
        have = min(k * dilation, xe - xs)
        idx = nn_.knn_batch_device(x[xs:xe].unsqueeze(0), y[ys:ye].unsqueeze(0), have)[0]     # [ny, have]
        keep = min(k, have)
        pick = torch.randint(have, (ye - ys, keep), dtype=torch.long, device=x.device, generator=generator)
        rows.append(torch.arange(ys, ye, device=x.device).unsqueeze(1).expand(-1, keep).reshape(-1))
        cols.append((idx.gather(1, pick) + xs).reshape(-1))
    return torch.cat(rows), torch.cat(cols)


def knn(x, y, k, batch_x=None, batch_y=None):
    """torch_cluster.knn: [2, E] = [y index; x index]."""
    row, col = _knn_segments(x, y, k, _segments(batch_x, x.shape[0]), _segments(batch_y, y.shape[0]))
    return torch.stack([row, col])


def knn_graph(pos, k, batch=None, loop=False):
    """torch_cluster.knn_graph (flow source_to_target): [2, E] = [neighbour j; node i]."""
    seg = _segments(batch, pos.shape[0])
    row, col = _knn_segments(pos, pos, k if loop else k + 1, seg, seg)
    if not loop:
        keep = row != col
        row, col = row[keep], col[keep]
    return torch.stack([col, row])


def radius_graph(pos, r, batch=None, loop=False, max_num_neighbors=32):
    """[2, E] = [neighbour j; node i] for |p_i - p_j| <= r, at most max_num_neighbors per node."""
    seg = _segments(batch, pos.shape[0])
    k = max_num_neighbors + (0 if loop else 1)
    row, col = _knn_segments(pos, pos, k, seg, seg)
    d2 = ((pos[row] - pos[col]) ** 2).sum(1)
    keep = d2 <= r * r
    if not loop:
        keep &= row != col
    return torch.stack([col[keep], row[keep]])


def fps(pos, batch=None, ratio=0.5, random_start=False):
    """Farthest point sampling per cloud -> sorted global indices (torch_cluster.fps).  One HIP workgroup per cloud
    (csrc/knn.hip: fps_kernel); the number of picks is ceil(ratio * n) per cloud."""
    import math

    from .. import _lib
    from ..graph import ptr, stream_ptr
    segs = _segments(batch, pos.shape[0])
    dev = pos.device
    counts = [e - s for s, e in segs]
    picks = [max(1, int(math.ceil(ratio * n))) if isinstance(ratio, float) else min(int(ratio), n) for n in counts]
    starts = torch.tensor([s for s, _ in segs], dtype=torch.int64, device=dev)
    cnt = torch.tensor(counts, dtype=torch.int64, device=dev)
    npick = torch.tensor(picks, dtype=torch.int64, device=dev)
    ostart = torch.cumsum(npick, 0) - npick
    first = (torch.stack([torch.randint(0, n, (1,)) for n in counts]).reshape(-1).to(dev) if random_start
             else torch.zeros(len(segs), dtype=torch.int64, device=dev))
    p = pos.detach().to(torch.float32).contiguous()
    out = torch.empty(sum(picks), dtype=torch.int64, device=dev)
    ws = torch.empty(pos.shape[0], dtype=torch.float32, device=dev)
    _lib.call('crfconv_fps', ptr(p), len(segs), ptr(starts), ptr(cnt), ptr(ostart), ptr(npick), ptr(first), ptr(ws),
              ptr(out), stream_ptr())
    pieces, o = [], 0
    for k in picks:
        pieces.append(out[o:o + k].sort().values)
        o += k
    return torch.cat(pieces)
