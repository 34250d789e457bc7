"""Dense PointConv encoder + CRF decoder -- drop-in for the reference network
(models/point_conv_big.py:8-167): same class names, constructor arguments, forward(data)
contract and state_dict keys, with the gather / weight-MLP / reduce chain fused into gfx950
kernels (crfconv_amd/csrc/pointconv.hip)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..graph import table_of
from .common import MLP, Base
from .continuous_crf_conv_big import ContinuousGaussianCRFConv as CRFConv


class PointConv(nn.Module):
    """Depth-wise point convolution: out_i = sum_k weight_nn(p_i - p_j) * x_j
    (models/point_conv_big.py:8-58).  `weight_nn` keeps the reference's module layout so
    checkpoints load, but is never materialised per edge."""

    def __init__(self, d_model):
        super(PointConv, self).__init__()
        self.weight_nn = nn.Sequential(MLP(3, d_model, activation=nn.LeakyReLU(negative_slope=0.1)),
                                       MLP(d_model, d_model, activation=None))

    def forward(self, x, pos, neighbor_idx):
        if torch.is_tensor(pos):
            src, tgt = pos, pos
        else:
            src, tgt = pos
        B, d = x.shape[0], x.shape[-1]
        table = table_of(neighbor_idx, src.shape[1])
        p_src = src.reshape(-1, 3)
        p_tgt = p_src if tgt is src else tgt.reshape(-1, 3)
        key = ('moments', p_src.data_ptr(), p_tgt.data_ptr())
        moments = table.cache.get(key)
        if moments is None:
            moments = ops.relpos_moments(p_src.float().contiguous(), p_tgt.float().contiguous(), table)
            table.cache[key] = moments
        l0, l1 = self.weight_nn[0], self.weight_nn[1]
        out = ops.point_conv(x.reshape(-1, d), p_src, p_tgt, table, l0.lin.weight, l0.bn.batch_norm,
                             l1.lin.weight, l1.bn.batch_norm, self.training, moments=moments)
        return out.reshape(B, -1, d)


class ResNetBBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super(ResNetBBlock, self).__init__()
        hidden_channels = out_channels // 4
        self.lin_in = MLP(in_channels, hidden_channels, activation=nn.LeakyReLU(negative_slope=0.1))
        self.lin_out = MLP(hidden_channels, out_channels, activation=None)
        if in_channels != out_channels:
            self.shortcut = MLP(in_channels, out_channels, activation=None)
        else:
            self.shortcut = nn.Identity()
        self.point_conv = PointConv(hidden_channels)

    @staticmethod
    def max_pooling(x, idx):
        B, C = x.shape[0], x.shape[-1]
        return ops.neighbor_maxpool(x.reshape(-1, C), table_of(idx, x.shape[1])).reshape(B, -1, C)

    def forward(self, x, pos, neighbor_idx):
        residual = self.shortcut(x)
        if not torch.is_tensor(pos):
            residual = self.max_pooling(residual, neighbor_idx)
        x = self.lin_in(x)
        x = self.point_conv(x, pos, neighbor_idx)
        x = self.lin_out(x)
        return F.leaky_relu(x + residual)


class Upsampling(nn.Module):
    def __init__(self, down_channels, up_channels, out_channels):
        super(Upsampling, self).__init__()
        self.lin = MLP(down_channels, up_channels, activation=nn.LeakyReLU(negative_slope=0.1))
        self.fusion = MLP(up_channels * 2, out_channels, activation=nn.LeakyReLU(negative_slope=0.1))

    @staticmethod
    def upsampling(x, idx):
        B, C = x.shape[0], x.shape[-1]
        return ops.gather_rows(x.reshape(-1, C), table_of(idx, x.shape[1])).reshape(B, -1, C)

    def forward(self, x_down, x_up, up_idx, neighbor_idx=None):
        x_down = self.upsampling(x_down, up_idx)
        x_down = self.lin(x_down)
        return self.fusion(torch.cat([x_up, x_down], dim=-1))


class PointConvResNet(Base):
    def __init__(self, in_channels, n_classes, use_crf=True, steps=1):
        super(PointConvResNet, self).__init__()
        layers = [32, 64, 128, 256, 512]
        self.C = n_classes
        self.conv1_1 = ResNetBBlock(in_channels, layers[0])
        self.conv1_2 = ResNetBBlock(layers[0], layers[0])
        self.conv2_1 = ResNetBBlock(layers[0], layers[1])
        self.conv2_2 = ResNetBBlock(layers[1], layers[1])
        self.conv3_1 = ResNetBBlock(layers[1], layers[2])
        self.conv3_2 = ResNetBBlock(layers[2], layers[2])
        self.conv4_1 = ResNetBBlock(layers[2], layers[3])
        self.conv4_2 = ResNetBBlock(layers[3], layers[3])
        self.conv5_1 = ResNetBBlock(layers[3], layers[4])
        self.conv5_2 = ResNetBBlock(layers[4], layers[4])

        def dec(down, up):
            return CRFConv(down, up, up, steps=steps) if use_crf else Upsampling(down, up, up)

        self.deconv4 = dec(layers[4], layers[3])
        self.deconv3 = dec(layers[3], layers[2])
        self.deconv2 = dec(layers[2], layers[1])
        self.deconv1 = dec(layers[1], layers[0])
        self.classifier = nn.Sequential(
            MLP(layers[0], layers[0] * 4, activation=nn.LeakyReLU(negative_slope=0.1)),
            nn.Dropout(p=0.5),
            nn.Linear(layers[0] * 4, n_classes))

    def forward(self, data):
        x, ms = data.x, data.multiscale
        x1 = self.conv1_1(x, ms[0].pos, ms[0].neighbor_idx)
        x1 = self.conv1_2(x1, ms[0].pos, ms[0].neighbor_idx)
        x2 = self.conv2_1(x1, (ms[0].pos, ms[1].pos), ms[0].sub_idx)
        x2 = self.conv2_2(x2, ms[1].pos, ms[1].neighbor_idx)
        x3 = self.conv3_1(x2, (ms[1].pos, ms[2].pos), ms[1].sub_idx)
        x3 = self.conv3_2(x3, ms[2].pos, ms[2].neighbor_idx)
        x4 = self.conv4_1(x3, (ms[2].pos, ms[3].pos), ms[2].sub_idx)
        x4 = self.conv4_2(x4, ms[3].pos, ms[3].neighbor_idx)
        x = self.conv5_1(x4, (ms[3].pos, ms[4].pos), ms[3].sub_idx)
        x = self.conv5_2(x, ms[4].pos, ms[4].neighbor_idx)
        x = self.deconv4(x, x4, ms[3].up_idx, ms[3].neighbor_idx)
        x = self.deconv3(x, x3, ms[2].up_idx, ms[2].neighbor_idx)
        x = self.deconv2(x, x2, ms[1].up_idx, ms[1].neighbor_idx)
        x = self.deconv1(x, x1, ms[0].up_idx, ms[0].neighbor_idx)
        x = self.classifier[1](self.classifier[0](x))
        x = ops.linear(x, self.classifier[2].weight, self.classifier[2].bias)
        return x.reshape(-1, self.C)
