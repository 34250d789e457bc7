"""Building blocks with the reference's names, signatures and state_dict layout
(models/common.py:26-40, 89-97).  FastBatchNorm1d mirrors the torch_points3d module the
reference imports (BatchNorm1d kept as ``.batch_norm``, statistics over every leading dim)."""
import torch
import torch.nn as nn

from .. import ops
from ..graph import require_gpu


class FastBatchNorm1d(nn.Module):
    def __init__(self, num_features, momentum=0.1, **kwargs):
        super().__init__()
        self.batch_norm = nn.BatchNorm1d(num_features, momentum=momentum, **kwargs)

    def forward(self, x):
        if x.dim() not in (2, 3):
            raise ValueError('Non supported number of dimensions {}'.format(x.dim()))
        bn = self.batch_norm
        if not bn.affine or x.dtype != torch.float32:
            raise ops._lib.CrfConvError('FastBatchNorm1d: affine float32 BatchNorm only (the kernels of csrc/bn.hip); got affine=%s, %s'
                                        % (bn.affine, x.dtype))
        return ops.bn_act(x, bn, self.training)          # statistics over every leading dim, one stats + one apply pass (csrc/bn.hip)


class MLP(nn.Module):
    """Linear(bias = not bn) -> BatchNorm -> activation (models/common.py:26-40)."""

    def __init__(self, in_channels, out_channels, bn=True, activation=None):
        super(MLP, self).__init__()
        self.lin = nn.Linear(in_channels, out_channels, bias=not bn)
        self.bn = FastBatchNorm1d(out_channels) if bn else None
        self.activation = activation

    def forward(self, x, *args, **kwargs):
        require_gpu(x)                                    # every layer of the path runs on the device; no CPU path
        co = self.lin.out_features
        fusable = (self.bn is not None and x.dtype == torch.float32 and co % 4 == 0 and co <= 1024
                   and self.bn.batch_norm.affine
                   and (self.activation is None or isinstance(self.activation, nn.LeakyReLU)))
        if fusable:      # Linear (MFMA, BatchNorm statistics in its epilogue) -> BatchNorm + LeakyReLU in one pass
            slope = 1.0 if self.activation is None else self.activation.negative_slope
            if ops.mlp_block_ok(x, self.lin.weight, self.lin.bias, self.bn.batch_norm, self.training):
                return ops.mlp_block(x, self.lin.weight, self.bn.batch_norm, slope)      # one autograd node, fused backward
            records = None
            if self.training:
                x, records = ops.linear(x, self.lin.weight, self.lin.bias, want_stats=True)
            else:
                x = ops.linear(x, self.lin.weight, self.lin.bias)
            slope = 1.0 if self.activation is None else self.activation.negative_slope
            return ops.bn_act(x, self.bn.batch_norm, self.training, slope, records=records)
        x = ops.linear(x, self.lin.weight, self.lin.bias)
        if self.bn is not None:
            x = self.bn(x)                                # FastBatchNorm1d: ops.bn_act (raises for what csrc/bn.hip does not take)
        if self.activation is not None:
            x = self.activation(x)                        # an activation module other than LeakyReLU is the caller's own
        return x


def mlp_fork(mlp, x):
    """(mlp(x), x') for an input with a second consumer (the ResNet block: lin_in(x) and shortcut(x),
    models/point_conv_big.py:83-88).  x' is x; where the fused block applies it is an alias whose gradient is added inside the
    block's backward while dX is written, instead of by an accumulation pass of autograd's (ops._MLPBlock, fork)."""
    if (mlp.training and mlp.bn is not None and x.dtype == torch.float32 and mlp.bn.batch_norm.affine and x.requires_grad
            and mlp.lin.out_features % 4 == 0 and mlp.lin.out_features <= 1024
            and (mlp.activation is None or isinstance(mlp.activation, nn.LeakyReLU))
            and ops.mlp_block_ok(x, mlp.lin.weight, mlp.lin.bias, mlp.bn.batch_norm, True)):
        require_gpu(x)
        slope = 1.0 if mlp.activation is None else mlp.activation.negative_slope
        return ops.mlp_block(x, mlp.lin.weight, mlp.bn.batch_norm, slope, fork=True)
    return mlp(x), x


def _group_spec(mlp, x):
    """(W, BatchNorm1d, slope) of an MLP that the fused training-mode blocks take -- Linear without bias, affine BatchNorm, no or a
    LeakyReLU activation --, else None."""
    if not (isinstance(mlp, MLP) and mlp.training and mlp.bn is not None and mlp.lin.bias is None and x.dtype == torch.float32
            and mlp.bn.batch_norm.affine and (mlp.activation is None or isinstance(mlp.activation, nn.LeakyReLU))):
        return None
    return mlp.lin.weight, mlp.bn.batch_norm, (1.0 if mlp.activation is None else mlp.activation.negative_slope)


def mlp_group(pairs, shared=False):
    """[(mlp, x, fork)] -> [mlp(x) or (mlp(x), x_alias)] with all the MLPs in ONE autograd node of two launches each way
    (ops.mlp_group: coarse-level training blocks whose inputs are all ready), or None when that form does not apply to every member
    -- the caller then runs them one by one."""
    blocks = []
    for mlp, x, fork in pairs:
        spec = _group_spec(mlp, x)
        if spec is None:
            return None
        require_gpu(x)
        blocks.append((x, spec[0], spec[1], spec[2], fork))
    return ops.mlp_group(blocks, shared=shared)


def mlp_join(mlp, x, skip, slope=0.01):
    """leaky_relu(mlp(x) + skip, slope) for an MLP without activation -- the tail of a ResNet block
    (models/point_conv_big.py:84-88).  One fused node (BatchNorm + add + LeakyReLU in a single pass) where it applies
    (training, MFMA-sized rows), else the module followed by ops.add_lrelu."""
    if (mlp.training and mlp.bn is not None and mlp.activation is None and mlp.lin.bias is None and x.dtype == torch.float32
            and mlp.bn.batch_norm.affine and mlp.lin.out_features % 4 == 0):
        out = ops.mlp_block_join(x, mlp.lin.weight, mlp.bn.batch_norm, skip, slope)
        if out is not None:
            return out
    return ops.add_lrelu(mlp(x), skip, slope)


class Base(nn.Module):
    """state_dict round trip (models/common.py:89-97)."""

    def save(self, filename):
        torch.save(self.state_dict(), filename)

    # A self-capturing model (train.autograph_forward) keeps its private graph runner in its instance dictionary, outside the module
    # tree.  The runner's graphs hold device addresses: a copy of the model starts without it, and whatever moves or casts the
    # parameters (to(), float(), cuda()) drops it -- the next training call captures again.
    def __getstate__(self):
        state = dict(super().__getstate__())
        state.pop('_autograph', None)
        return state

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop('_autograph', None)
        return super()._apply(fn, *args, **kwargs)

    def load(self, filename):
        self.load_state_dict(torch.load(filename))
