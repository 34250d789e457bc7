"""Dense continuous-Gaussian-CRF convolution on gfx950 kernels.

Drop-in for the reference layer (models/continuous_crf_conv_big.py:7-78): constructor arguments, the
forward signature, parameter names and shapes are the reference's.  What differs is how it runs: the
feature-space Gaussian softmax and all mean-field steps are the fused kernels of csrc/crf.hip, the
nearest up-sampling is a row gather, and (I + C)^-1 is computed once per call instead of once per step."""
import torch
import torch.nn as nn

from .. import ops
from ..graph import table_of
from .common import MLP, mlp_fork, mlp_group


def _embed(cin, hidden):
    """Two-layer per-point embedding: MLP(cin -> hidden, LeakyReLU 0.1) then MLP(hidden -> hidden)."""
    return nn.Sequential(MLP(cin, hidden, activation=nn.LeakyReLU(negative_slope=0.1)), MLP(hidden, hidden, activation=None))


class ContinuousGaussianCRFConv(nn.Module):
    def __init__(self, unary_channels, pairwise_channels, out_channels=None, steps=1):
        super().__init__()
        self.unary_channels, self.pairwise_channels = unary_channels, pairwise_channels
        self.out_channels = pairwise_channels if out_channels is None else out_channels
        self.hidden_channels = self.out_channels // 4
        self.steps = steps
        H, O = self.hidden_channels, self.out_channels
        self.unary_nn = _embed(unary_channels, H)            # on the COARSE level's features
        self.pairwise_nn = _embed(pairwise_channels, H)      # on this level's encoder (skip) features
        self.out_nn = MLP(H, O, activation=nn.LeakyReLU(negative_slope=0.1))
        self.fusion_nn = MLP(2 * O, O, activation=nn.LeakyReLU(negative_slope=0.1))
        self.c = nn.Parameter(torch.empty(H, H))             # label compatibility factor, C = c^T c
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.eye_(self.c)

    def forward(self, unary, pairwise, up_idx, neighbor_idx, matrices=None):
        """unary [B, N', U] (coarse level), pairwise [B, N, P], up_idx [B, N, 1] -> nearest coarse point,
        neighbor_idx [B, N, K] whose column 0 is the query itself.  Returns [B, N, O].  `matrices`: this layer's
        (Q, P) = ((I + c^T c)^-1, I - Q) when the network has computed all its layers' in one launch."""
        B, N, _ = pairwise.shape
        H = self.hidden_channels
        # unary_nn (coarse rows) and pairwise_nn (this level's rows) are independent chains of two blocks: where both are coarse-level
        # launches (latency chains on a fraction of the chip) block i of the one runs beside block i of the other (common.mlp_group)
        first = mlp_group([(self.unary_nn[0], unary, False), (self.pairwise_nn[0], pairwise, True)])
        second = None
        if first is not None:
            c0, (g0, pairwise) = first
            second = mlp_group([(self.unary_nn[1], c0, False), (self.pairwise_nn[1], g0, False)])
            if second is None:
                coarse, guide = self.unary_nn[1](c0).reshape(-1, H), self.pairwise_nn[1](g0).reshape(-1, H)
            else:
                coarse, guide = second[0].reshape(-1, H), second[1].reshape(-1, H)
        else:
            coarse = self.unary_nn(unary).reshape(-1, H)
            # `pairwise` feeds pairwise_nn and fusion_nn: the second reads the alias the first hands on (common.mlp_fork)
            guide, pairwise = mlp_fork(self.pairwise_nn[0], pairwise)
            guide = self.pairwise_nn[1](guide).reshape(-1, H)
        z = ops.gather_rows(coarse, table_of(up_idx, unary.shape[1]))                      # up-sample the unary term
        field = ops.crf_meanfield(z, guide, self.c, table_of(neighbor_idx, N), self.steps, k0=1,      # k0 = 1: no self edge
                                  matrices=matrices)
        refined = self.out_nn(field.reshape(B, N, H))
        fus = self.fusion_nn
        if fus.bn is not None and fus.lin.bias is None and isinstance(fus.activation, nn.LeakyReLU):
            # fusion_nn(cat[refined, pairwise]) with the concatenation left implicit (two operand pointers)
            out = ops.mlp_block_cat(refined, pairwise, fus.lin.weight, fus.bn.batch_norm, fus.training,
                                    fus.activation.negative_slope)
            if out is not None:
                return out
        return fus(ops.cat2(refined, pairwise))
