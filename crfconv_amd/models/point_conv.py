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
            residual = self.mlp4(residual)
        h = self.mlp2(x)
        msg = ops.point_conv(h, pos_src, pos_dst, table, self.mlp1[0].weight, self.mlp1[1], self.mlp1[3].weight,
                             self.mlp1[4], self.training, slope=self.mlp1[2].negative_slope)
        return F.leaky_relu(self.mlp3(msg) + residual)


def build_graph(pos, batch, method='radius', r=0.1, k=16, dilation=1, loop=True):
    """models/point_conv.py:341-366."""
    assert method in ['radius', 'knn']
    if method == 'radius':
        return graph_ops.radius_graph(pos, r, batch, loop=loop, max_num_neighbors=k)
    edge_index = graph_ops.knn_graph(pos, k * dilation, batch, loop=loop)
    if dilation > 1:
        n = pos.shape[0]
        index = torch.randint(k * dilation, (n, k), dtype=torch.long, device=edge_index.device)
        arange = torch.arange(n, dtype=torch.long, device=edge_index.device) * (k * dilation)
        edge_index = edge_index[:, (index + arange.view(-1, 1)).view(-1)]
    return edge_index


def build_bipartite_graph(pos, batch, ratio, method='radius', r=0.1, k=32, dilation=1):
    """models/point_conv.py:368-396: fps-sampled targets, each gathering from the full cloud."""
    assert method in ['radius', 'knn']
    idx = graph_ops.fps(pos, batch, ratio=ratio)
    sub_pos, sub_batch = pos[idx], (batch[idx] if batch is not None else None)
    row, col = graph_ops.knn(pos, sub_pos, k * dilation, batch, sub_batch)
    if method == 'radius':
        keep = ((sub_pos[row] - pos[col]) ** 2).sum(1) <= r * r
        row, col = row[keep], col[keep]
    elif dilation > 1:
        n = idx.shape[0]
        index = torch.randint(k * dilation, (n, k), dtype=torch.long, device=row.device)
        arange = torch.arange(n, dtype=torch.long, device=row.device) * (k * dilation)
        sel = (index + arange.view(-1, 1)).view(-1)
        row, col = row[sel], col[sel]
    return torch.stack([col, row], dim=0), sub_pos, sub_batch
