"""Sparse (edge-list) depth-wise separable point convolution -- drop-in for
`DepthwiseSeparablePointConv` (models/point_conv.py:13-66): same constructor, forward(x, pos, edge_index)
and parameter names (mlp1 .. mlp4), on the fused PointConv kernels through a padded neighbour table.
Graph builders with the reference's names (`build_graph`, `build_bipartite_graph`, :341-396)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..graph import table_from_edges
from . import graph_ops
from .discrete_crf_conv import DiscreteCRFConv


class DepthwiseSeparablePointConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super(DepthwiseSeparablePointConv, self).__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.hidden_channels = out_channels // 4
        h = self.hidden_channels
        self.mlp1 = nn.Sequential(nn.Linear(3, h, bias=False), nn.BatchNorm1d(h), nn.LeakyReLU(inplace=True),
                                  nn.Linear(h, h, bias=False), nn.BatchNorm1d(h))
        self.mlp2 = nn.Sequential(nn.Linear(self.in_channels, h, bias=False), nn.BatchNorm1d(h), nn.LeakyReLU(inplace=True))
        self.mlp3 = nn.Sequential(nn.Linear(h, self.out_channels, bias=False), nn.BatchNorm1d(self.out_channels))
        if self.in_channels != self.out_channels:
            self.mlp4 = nn.Sequential(nn.Linear(self.in_channels, self.out_channels), nn.BatchNorm1d(self.out_channels))

    def forward(self, x, pos, edge_index):
        """edge_index = [col (source j); row (target i)].  Symmetric graphs (tensor `pos`) get their self loops
        removed and re-added (point_conv.py:45-47); bipartite graphs take pos = (pos_src, pos_dst)."""
        bipartite = not torch.is_tensor(pos)
        if not bipartite:
            n = pos.shape[0]
            keep = edge_index[0] != edge_index[1]
            loops = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
            edge_index = torch.cat([edge_index[:, keep], torch.stack([loops, loops])], dim=1)
            pos_src, pos_dst = pos, pos
        else:
            pos_src, pos_dst = pos
        col, row = edge_index
        table = table_from_edges(row, col, pos_dst.shape[0], pos_src.shape[0])
        residual = x
        if bipartite:
            residual = ops.neighbor_maxpool(residual, table)          # scatter_max(x[col], row)
        if self.in_channels != self.out_channels:
            residual = ops.run_lin_bn(self.mlp4, residual)
        h = ops.run_lin_bn(self.mlp2, x)
        msg = ops.point_conv(h, pos_src, pos_dst, table, self.mlp1[0].weight, self.mlp1[1], self.mlp1[3].weight,
                             self.mlp1[4], self.training, slope=self.mlp1[2].negative_slope)
        return ops.add_lrelu(ops.run_lin_bn(self.mlp3, msg), residual, 0.01)      # F.leaky_relu default slope


def build_graph(pos, batch, method='radius', r=0.1, k=16, dilation=1, loop=True):
    """models/point_conv.py:341-366."""
    assert method in ['radius', 'knn']
    if method == 'radius':
        return graph_ops.radius_graph(pos, r, batch, loop=loop, max_num_neighbors=k)
    if dilation > 1:
        row, col = graph_ops.knn_dilated(pos, pos, k, dilation, batch, batch)
        if not loop:
            keep = row != col
            row, col = row[keep], col[keep]
        return torch.stack([col, row])
    return graph_ops.knn_graph(pos, k, batch, loop=loop)


def build_bipartite_graph(pos, batch, ratio, method='radius', r=0.1, k=32, dilation=1):
    """models/point_conv.py:368-396: fps-sampled targets, each gathering from the full cloud."""
    assert method in ['radius', 'knn']
    idx = graph_ops.fps(pos, batch, ratio=ratio)
    sub_pos, sub_batch = pos[idx], (batch[idx] if batch is not None else None)
    if method == 'radius':
        row, col = graph_ops.knn(pos, sub_pos, k * dilation, batch, sub_batch)
        keep = ((sub_pos[row] - pos[col]) ** 2).sum(1) <= r * r
        row, col = row[keep], col[keep]
    elif dilation > 1:
        row, col = graph_ops.knn_dilated(pos, sub_pos, k, dilation, batch, sub_batch)
    else:
        row, col = graph_ops.knn(pos, sub_pos, k, batch, sub_batch)
    return torch.stack([col, row], dim=0), sub_pos, sub_batch


# ---------------------------------------------------------------------------------------------------- networks
# The sparse segmentation networks of models/point_conv.py:69-618, assembled from the operators above.  In the
# reference they cannot be constructed (`DSPointConv` / `knn_interpolate` are undefined there); the
# structure, parameter names and forward(data) contracts below follow what its source spells out.
ENC_WIDTHS = (32, 64, 128, 256, 512)


def knn_interpolate(x, pos_x, pos_y, batch_x=None, batch_y=None, k=3):
    """torch_geometric.nn.knn_interpolate: inverse-squared-distance blend of the k nearest source features."""
    row, col = graph_ops.knn(pos_x, pos_y, k, batch_x, batch_y)                 # y index, x index
    w = 1.0 / ((pos_y[row] - pos_x[col]) ** 2).sum(1, keepdim=True).clamp_min(1e-16)
    num = torch.zeros((pos_y.shape[0], x.shape[1]), dtype=x.dtype, device=x.device).index_add_(0, row, x[col] * w)
    den = torch.zeros((pos_y.shape[0], 1), dtype=x.dtype, device=x.device).index_add_(0, row, w)
    return num / den


def _lin_bn_act(cin, cout):
    return nn.Sequential(nn.Linear(cin, cout), nn.BatchNorm1d(cout), nn.LeakyReLU(inplace=True))


class _SparseEncoder(nn.Module):
    """conv{l}_1 / conv{l}_2 pairs over an fps pyramid (point_conv.py:84-98, 302-315, 197-264)."""

    def __init__(self, in_channels, method, ratio, radius, kernel_size, dilation):
        super().__init__()
        assert method in ['radius', 'knn']
        self.method, self.ratio, self.radius = method, ratio, radius
        self.kernel_size, self.dilation = kernel_size, dilation
        cin = in_channels
        for lvl, width in enumerate(ENC_WIDTHS, start=1):
            setattr(self, 'conv%d_1' % lvl, DepthwiseSeparablePointConv(cin, width))
            setattr(self, 'conv%d_2' % lvl, DepthwiseSeparablePointConv(width, width))
            cin = width

    def build_graph(self, pos, batch, method='radius', r=0.1, k=16, dilation=1, loop=True):
        return build_graph(pos, batch, method=method, r=r, k=k, dilation=dilation, loop=loop)

    def build_bipartite_graph(self, pos, batch, ratio, method='radius', r=0.1, k=32, dilation=1):
        return build_bipartite_graph(pos, batch, ratio, method=method, r=r, k=k, dilation=dilation)

    def encode(self, x, pos, batch):
        """-> per level (features, pos, batch), finest first."""
        levels = []
        for lvl in range(len(ENC_WIDTHS)):
            if lvl > 0:
                ei, pos_c, batch_c = self.build_bipartite_graph(pos, batch, self.ratio[lvl - 1], method=self.method,
                                                                r=self.radius[lvl - 1], k=self.kernel_size[lvl - 1],
                                                                dilation=self.dilation[lvl - 1])
                x = getattr(self, 'conv%d_1' % (lvl + 1))(x, (pos, pos_c), ei)
                pos, batch = pos_c, batch_c
            ei = self.build_graph(pos, batch, method=self.method, r=self.radius[lvl], k=self.kernel_size[lvl],
                                  dilation=self.dilation[lvl])
            if lvl == 0:
                x = self.conv1_1(x, pos, ei)
            x = getattr(self, 'conv%d_2' % (lvl + 1))(x, pos, ei)
            levels.append((x, pos, batch))
        return levels


class Baseline(_SparseEncoder):
    """Encoder + interpolation decoder (point_conv.py:69-282)."""

    def __init__(self, in_channels, method='radius', ratio=None, radius=None, kernel_size=16, dilation=None):
        super().__init__(in_channels, method, ratio, radius, kernel_size, dilation)
        for lvl in range(4, 0, -1):
            setattr(self, 'lin%d' % lvl, _lin_bn_act(ENC_WIDTHS[lvl], ENC_WIDTHS[lvl - 1]))
            if lvl < 4:
                setattr(self, 'fusion%d' % lvl, _lin_bn_act(2 * ENC_WIDTHS[lvl], ENC_WIDTHS[lvl]))

    def forward(self, x, pos, batch):
        levels = self.encode(x, pos, batch)
        h = levels[4][0]
        for lvl in range(3, -1, -1):
            if lvl < 3:
                h = ops.run_lin_bn(getattr(self, 'fusion%d' % (lvl + 1)), torch.cat([h, levels[lvl + 1][0]], dim=1))
            h = knn_interpolate(h, levels[lvl + 1][1], levels[lvl][1], levels[lvl + 1][2], levels[lvl][2], k=3)
            h = ops.run_lin_bn(getattr(self, 'lin%d' % (lvl + 1)), h)
        return torch.cat([h, levels[0][0]], dim=1)


class PointConvGassuianCRFNet(_SparseEncoder):
    """Encoder + guided Gaussian-CRF decoder (point_conv.py:285-483)."""

    def __init__(self, in_channels, method='radius', ratio=None, radius=None, kernel_size=16, dilation=None, steps=1):
        super().__init__(in_channels, method, ratio, radius, kernel_size, dilation)
        from .continuous_crf_conv import GuideGaussianCRFConv as GCRFConv
        for lvl in range(4, 0, -1):
            setattr(self, 'deconv%d' % lvl, GCRFConv(ENC_WIDTHS[lvl], ENC_WIDTHS[lvl - 1], radius=self.radius[lvl - 1],
                                                     kernel_size=self.kernel_size[lvl - 1], steps=steps))
            if lvl < 4:
                setattr(self, 'fusion%d' % lvl, _lin_bn_act(2 * ENC_WIDTHS[lvl], ENC_WIDTHS[lvl]))

    def forward(self, x, pos, batch):
        levels = self.encode(x, pos, batch)
        h = levels[4][0]
        for lvl in range(3, -1, -1):
            if lvl < 3:
                h = ops.run_lin_bn(getattr(self, 'fusion%d' % (lvl + 1)), torch.cat([h, levels[lvl + 1][0]], dim=1))
            h = knn_interpolate(h, levels[lvl + 1][1], levels[lvl][1], levels[lvl + 1][2], levels[lvl][2], k=3)
            feat, p, b = levels[lvl]
            h = getattr(self, 'deconv%d' % (lvl + 1))(h, feat, p, b)
        return torch.cat([h, levels[0][0]], dim=1)


def _head(cin, hidden, n_classes):
    return nn.Sequential(nn.Linear(cin, hidden), nn.ReLU(inplace=True), nn.Linear(hidden, n_classes))


def _run_head(head, x):
    """Linear -> ReLU -> Linear of the classifier heads on ops.linear (MFMA kernels for the per-point rows)."""
    return ops.linear(F.relu(ops.linear(x, head[0].weight, head[0].bias)), head[2].weight, head[2].bias)


class CRFSegNet(nn.Module):
    """point_conv.py:566-591: forward(data) reads data.pos / data.x / data.batch, returns log-probabilities."""

    def __init__(self, in_channels, n_classes=2, steps=1):
        super().__init__()
        self.feature = PointConvGassuianCRFNet(in_channels, method='knn', ratio=[0.25] * 4, radius=[0.2] * 5,
                                               kernel_size=[16] * 5, dilation=[1] * 5, steps=steps)
        self.classifier = _head(32 + 32, 128, n_classes)

    def forward(self, data):
        x = self.feature(x=data.x, pos=data.pos, batch=data.batch)
        return F.log_softmax(_run_head(self.classifier, x), dim=-1)


class BaselineSegNet(nn.Module):
    """point_conv.py:519-539."""

    def __init__(self, in_channels, n_classes=2):
        super().__init__()
        self.feature = Baseline(in_channels, method='knn', ratio=[0.25] * 4, radius=[0.2] * 5, kernel_size=[16] * 5,
                                dilation=[1] * 5)
        self.classifier = _head(32 + 32, 128, n_classes)

    def forward(self, data):
        x = self.feature(x=data.x, pos=data.pos, batch=data.batch)
        return F.log_softmax(_run_head(self.classifier, x), dim=-1)


class CRFSegNet_Part(nn.Module):
    """point_conv.py:491-512 (ShapeNet-Part: normals as extra input, one-hot object category at the classifier).
    The reference sizes its classifier for 64 + 64 + 16 inputs although its backbone emits 32 + 32 channels; the
    backbone width is what the forward pass produces, so the classifier here takes 32 + 32 + 16."""

    def __init__(self, in_channels, n_classes=2, steps=1):
        super().__init__()
        self.feature = PointConvGassuianCRFNet(in_channels, method='knn', ratio=[0.25, 0.5, 0.5, 0.5],
                                               radius=[0.2, 0.4, 0.6, 0.8, 1.0], kernel_size=[32, 16, 8, 8, 8],
                                               dilation=[1, 2, 4, 2, 1], steps=steps)
        self.classifier = _head(32 + 32 + 16, 256, n_classes)

    def forward(self, data):
        c = F.one_hot(data.category[data.batch], num_classes=16).float()
        x = self.feature(x=torch.cat([data.pos, data.norm], dim=1), pos=data.pos, batch=data.batch)
        return F.log_softmax(_run_head(self.classifier, torch.cat([x, c], dim=1)), dim=-1)


class BaselineDiscreteCRFSegNet(nn.Module):
    """point_conv.py:545-565: plain encoder, classifier probabilities refined by the label-space CRF; returns
    (log p, log q)."""

    def __init__(self, in_channels, n_classes=2, steps=1):
        super().__init__()
        self.feature = Baseline(in_channels, method='knn', ratio=[0.25, 0.375, 0.375, 0.375], radius=[0.2] * 5,
                                kernel_size=[32, 16, 16, 16, 16], dilation=[1, 2, 4, 4, 2])
        # the reference sizes this classifier for 64 + 64 inputs; its backbone emits 32 + 32 (same note as CRFSegNet_Part)
        self.classifier = _head(32 + 32, 256, n_classes)
        self.crf = DiscreteCRFConv(n_classes, in_channels, radius=0.2, kernel_size=32, steps=steps)

    def forward(self, data):
        p = torch.softmax(_run_head(self.classifier, self.feature(x=data.x, pos=data.pos, batch=data.batch)), dim=-1)
        q = self.crf(data.pos, p, f=data.x, batch=data.batch)
        return torch.log(p), torch.log(q)


class DualCRFSegNet(nn.Module):
    """point_conv.py:594-617: continuous-CRF encoder + label-space CRF on its class probabilities."""

    def __init__(self, in_channels, n_classes=2, steps=1):
        super().__init__()
        self.feature = PointConvGassuianCRFNet(in_channels, method='knn', ratio=[0.25, 0.375, 0.375, 0.375],
                                               radius=[0.2] * 5, kernel_size=[32, 16, 16, 16, 16],
                                               dilation=[1, 2, 4, 4, 2], steps=steps)
        self.classifier = _head(32 + 32, 256, n_classes)
        self.crf = DiscreteCRFConv(n_classes, in_channels, radius=0.2, kernel_size=32, steps=steps)

    def forward(self, data):
        p = torch.softmax(_run_head(self.classifier, self.feature(x=data.x, pos=data.pos, batch=data.batch)), dim=-1)
        q = self.crf(data.pos, p, f=data.x, batch=data.batch)
        return torch.log(p), torch.log(q)
