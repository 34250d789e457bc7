"""Edge-list (sparse) continuous-Gaussian-CRF layers on the gfx950 mean-field kernels.

Drop-ins for models/continuous_crf_conv.py:9-69 (`GuideGaussianCRFConv`) and :72-133
(`ContinuousGaussianCRFConv`): constructor arguments, forward signatures and parameter names are the
reference's.  The variable-degree graph is packed into a padded neighbour table
(graph.table_from_edges) and runs through the same kernels as the dense layer (csrc/crf.hip, generic path
with "no neighbour" entries)."""
import torch
import torch.nn as nn

from .. import ops
from ..graph import table_from_edges
from . import graph_ops


def _lin_bn(cin, cout, act=False):
    """Linear(no bias) -> BatchNorm1d [-> LeakyReLU(0.01)]: indices 0 / 1 / 2 as in the reference's Sequentials."""
    layers = [nn.Linear(cin, cout, bias=False), nn.BatchNorm1d(cout)]
    if act:
        layers.append(nn.LeakyReLU(inplace=True))
    return nn.Sequential(*layers)


def _identity_(param):
    nn.init.eye_(param)


class GuideGaussianCRFConv(nn.Module):
    def __init__(self, in_n_channels, in_e_channels, out_channels=None, radius=0.1, kernel_size=32, steps=1):
        super().__init__()
        self.in_n_channels, self.in_e_channels = in_n_channels, in_e_channels
        self.out_channels = in_e_channels if out_channels is None else out_channels
        self.radius, self.kernel_size, self.steps = radius, kernel_size, steps
        self.unary = _lin_bn(in_n_channels, self.out_channels)
        self.pairwise = _lin_bn(in_e_channels, self.out_channels, act=True)
        self.c = nn.Parameter(torch.empty(self.out_channels, self.out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        _identity_(self.c)

    def forward(self, x, y, pos, batch, edge_index=None):
        """x [N, Cn] unary input, y [N, Ce] guidance, pos [N, 3], batch [N].  The graph is the radius graph of
        `pos` (reference :53) unless `edge_index` ([2, E]: row 0 = source j, row 1 = target i) is supplied."""
        if edge_index is None:
            edge_index = graph_ops.radius_graph(pos, self.radius, batch, loop=False, max_num_neighbors=self.kernel_size)
        src, tgt = edge_index[0], edge_index[1]
        n = pos.shape[0]
        field = ops.crf_meanfield(ops.run_lin_bn(self.unary, x), ops.run_lin_bn(self.pairwise, y), self.c,
                                  table_from_edges(tgt, src, n, n), self.steps, k0=0)
        return ops.leaky_relu(field, 0.01)                  # F.leaky_relu's default slope (reference :69)


class ContinuousGaussianCRFConv(nn.Module):
    def __init__(self, unary_channels, pairwise_channels, hidden_channels=None, out_channels=None, steps=1):
        super().__init__()
        self.unary_channels, self.pairwise_channels = unary_channels, pairwise_channels
        self.out_channels = pairwise_channels if out_channels is None else out_channels
        self.hidden_channels = self.out_channels // 4 if hidden_channels is None else hidden_channels
        self.steps = steps
        self.unary_net = _lin_bn(unary_channels, self.hidden_channels)
        self.pairwise_net = _lin_bn(pairwise_channels, self.hidden_channels)
        self.mlp = _lin_bn(self.hidden_channels, self.out_channels, act=True)
        self.fusion_net = _lin_bn(2 * self.out_channels, self.out_channels, act=True)
        self.c = nn.Parameter(torch.empty(self.hidden_channels, self.hidden_channels))
        self._reset_parameters()

    def _reset_parameters(self):
        _identity_(self.c)

    def forward(self, x, y, pos, edge_index):
        """Node i = edge_index[0] aggregates from j = edge_index[1] (reference :114, 126)."""
        n = pos.shape[0]
        table = table_from_edges(edge_index[0], edge_index[1], n, n)
        field = ops.crf_meanfield(ops.run_lin_bn(self.unary_net, x), ops.run_lin_bn(self.pairwise_net, y), self.c, table,
                                  self.steps, k0=0)
        return ops.run_lin_bn(self.fusion_net, ops.cat2(ops.run_lin_bn(self.mlp, field), y))
