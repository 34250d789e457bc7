"""Dense continuous-Gaussian-CRF convolution -- drop-in for the reference module
(models/continuous_crf_conv_big.py:7-78): same constructor, forward signature, parameter names
and shapes; the similarity softmax and the mean-field loop run as fused gfx950 kernels."""
import torch
import torch.nn as nn

from .. import ops
from ..graph import table_of
from .common import MLP


class ContinuousGaussianCRFConv(nn.Module):
    def __init__(self, unary_channels, pairwise_channels, out_channels=None, steps=1):
        super(ContinuousGaussianCRFConv, self).__init__()
        self.unary_channels = unary_channels
        self.pairwise_channels = pairwise_channels
        self.out_channels = out_channels if out_channels is not None else pairwise_channels
        self.hidden_channels = self.out_channels // 4
        self.steps = steps

        def act():
            return nn.LeakyReLU(negative_slope=0.1)

        self.unary_nn = nn.Sequential(MLP(self.unary_channels, self.hidden_channels, activation=act()),
                                      MLP(self.hidden_channels, self.hidden_channels, activation=None))
        self.pairwise_nn = nn.Sequential(MLP(self.pairwise_channels, self.hidden_channels, activation=act()),
                                         MLP(self.hidden_channels, self.hidden_channels, activation=None))
        self.out_nn = MLP(self.hidden_channels, self.out_channels, activation=act())
        self.fusion_nn = MLP(self.out_channels * 2, self.out_channels, activation=act())
        self.c = nn.Parameter(torch.Tensor(self.hidden_channels, self.hidden_channels))
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.eye_(self.c)

    def forward(self, unary, pairwise, up_idx, neighbor_idx):
        """unary [B, N', U] (coarse level), pairwise [B, N, P], up_idx [B, N, 1] into the coarse
        level, neighbor_idx [B, N, K] with column 0 the query itself  ->  [B, N, O]."""
        B, N = pairwise.shape[0], pairwise.shape[1]
        H = self.hidden_channels
        nbr = table_of(neighbor_idx, N)
        up = table_of(up_idx, unary.shape[1])
        x = self.unary_nn(unary).reshape(-1, H)
        y = self.pairwise_nn(pairwise).reshape(-1, H)
        z = ops.gather_rows(x, up)                                  # nearest-coarse-point up-sampling
        x = ops.crf_meanfield(z, y, self.c, nbr, self.steps, k0=1)  # k0 = 1: drop the self column
        x = self.out_nn(x.reshape(B, N, H))
        return self.fusion_nn(torch.cat([x, pairwise], dim=-1))
