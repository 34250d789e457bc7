"""Sparse (edge-list) continuous-Gaussian-CRF layers -- drop-ins for
models/continuous_crf_conv.py:9-69 (`GuideGaussianCRFConv`) and :72-133 (`ContinuousGaussianCRFConv`):
same constructors, forward signatures and parameter names.  The variable-degree graph is packed into a
padded neighbour table (graph.table_from_edges) and runs on the same mean-field kernels as the dense
layer (csrc/crf.hip, generic path with "no neighbour" entries)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..graph import table_from_edges
from . import graph_ops


class GuideGaussianCRFConv(nn.Module):
    def __init__(self, in_n_channels, in_e_channels, out_channels=None, radius=0.1, kernel_size=32, steps=1):
        super(GuideGaussianCRFConv, self).__init__()
        self.in_n_channels = in_n_channels
        self.in_e_channels = in_e_channels
        self.out_channels = out_channels if out_channels is not None else in_e_channels
        self.radius = radius
        self.kernel_size = kernel_size
        self.steps = steps
        self.unary = nn.Sequential(nn.Linear(self.in_n_channels, self.out_channels, bias=False),
                                   nn.BatchNorm1d(self.out_channels))
        self.pairwise = nn.Sequential(nn.Linear(self.in_e_channels, self.out_channels, bias=False),
                                      nn.BatchNorm1d(self.out_channels), nn.LeakyReLU(inplace=True))
        self.c = nn.Parameter(torch.Tensor(self.out_channels, self.out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.eye_(self.c)

    def forward(self, x, y, pos, batch, edge_index=None):
        """x [N, Cn], y [N, Ce], pos [N, 3], batch [N].  `edge_index` ([2, E], row 0 = source j, row 1 = target i)
        overrides the radius graph the reference builds at continuous_crf_conv.py:53."""
        N = pos.shape[0]
        if edge_index is None:
            edge_index = graph_ops.radius_graph(pos, self.radius, batch, loop=False, max_num_neighbors=self.kernel_size)
        col, row = edge_index
        x = self.unary(x)
        y = self.pairwise(y)
        table = table_from_edges(row, col, N, N)
        x = ops.crf_meanfield(x, y, self.c, table, self.steps, k0=0)
        return F.leaky_relu(x)


class ContinuousGaussianCRFConv(nn.Module):
    def __init__(self, unary_channels, pairwise_channels, hidden_channels=None, out_channels=None, steps=1):
        super(ContinuousGaussianCRFConv, self).__init__()
        self.unary_channels = unary_channels
        self.pairwise_channels = pairwise_channels
        self.out_channels = out_channels if out_channels is not None else pairwise_channels
        self.hidden_channels = hidden_channels if hidden_channels is not None else self.out_channels // 4
        self.steps = steps
        H, O = self.hidden_channels, self.out_channels
        self.unary_net = nn.Sequential(nn.Linear(self.unary_channels, H, bias=False), nn.BatchNorm1d(H))
        self.pairwise_net = nn.Sequential(nn.Linear(self.pairwise_channels, H, bias=False), nn.BatchNorm1d(H))
        self.mlp = nn.Sequential(nn.Linear(H, O, bias=False), nn.BatchNorm1d(O), nn.LeakyReLU(inplace=True))
        self.fusion_net = nn.Sequential(nn.Linear(O * 2, O, bias=False), nn.BatchNorm1d(O), nn.LeakyReLU(inplace=True))
        self.c = nn.Parameter(torch.Tensor(H, H))
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.eye_(self.c)

    def forward(self, x, y, pos, edge_index):
        """Messages flow j = edge_index[1] -> i = edge_index[0] (continuous_crf_conv.py:114,126)."""
        N = pos.shape[0]
        i, j = edge_index
        xh = self.unary_net(x)
        s = self.pairwise_net(y)
        table = table_from_edges(i, j, N, N)
        xh = ops.crf_meanfield(xh, s, self.c, table, self.steps, k0=0)
        xh = self.mlp(xh)
        return self.fusion_net(torch.cat([xh, y], dim=-1))
