"""Label-space (discrete) CRF layer on the gfx950 kernels -- drop-in for models/discrete_crf_conv.py:11-63.

Same constructor, parameter names (``F`` [K, D, H], ``W`` [K, 1], ``C`` [L, L]) and initialisation; ``forward(pos,
p, f, batch)`` returns the refined label distribution q [N, L].  The radius graph comes from the device graph
builder; the Gaussian-kernel edge weights and the mean-field steps run in csrc/discrete.hip / csrc/crf.hip."""
import torch
import torch.nn as nn

from .. import ops
from ..graph import table_from_edges
from . import graph_ops


class DiscreteCRFConv(nn.Module):
    def __init__(self, n_channels, e_channels, hidden_channels=64, num_kernels=5, radius=0.2, kernel_size=32, steps=5):
        super().__init__()
        self.n_channels, self.e_channels, self.hidden_channels = n_channels, e_channels, hidden_channels
        self.radius, self.kernel_size, self.num_kernels, self.steps = radius, kernel_size, num_kernels, steps
        self.F = nn.Parameter(torch.empty(num_kernels, e_channels, hidden_channels))
        self.W = nn.Parameter(torch.empty(num_kernels, 1))
        self.C = nn.Parameter(torch.empty(n_channels, n_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.uniform_(self.F)
        nn.init.constant_(self.W, 1 / self.num_kernels)
        nn.init.eye_(self.C)

    def forward(self, pos, p, f=None, batch=None, edge_index=None):
        """pos [N, 3]; p [N, L] label probabilities; f [N, D] features the kernels act on; batch [N].  The graph is
        the radius graph of `pos` (:44) unless `edge_index` ([2, E]: row 0 = source ``col``, row 1 = target ``row``)
        is supplied."""
        n = pos.shape[0]
        if edge_index is None:
            edge_index = graph_ops.radius_graph(pos, self.radius, batch, loop=False, max_num_neighbors=self.kernel_size)
        table = table_from_edges(edge_index[1], edge_index[0], n, n)
        G, D, H = self.F.shape
        fk = ops.linear(f, self.F.permute(0, 2, 1).reshape(G * H, D))  # f F_g for every kernel g: [N, G * H], on the library's products
        w = ops.kernel_weights(fk, self.W.reshape(-1), table, G, H)   # [N, Kp], 0 on missing entries
        return ops.discrete_meanfield(p, -torch.log(p), w, self.C, table, self.steps)
