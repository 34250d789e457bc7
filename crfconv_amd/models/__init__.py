"""Same export surface as the reference's models/__init__.py:1-2 for the dense path."""
from .common import MLP, Base, FastBatchNorm1d
from .continuous_crf_conv_big import ContinuousGaussianCRFConv
from .point_conv_big import PointConv, PointConvResNet, ResNetBBlock, Upsampling
from .point_conv_big import PointConvResNet as PointConvBig

__all__ = ['MLP', 'Base', 'FastBatchNorm1d', 'ContinuousGaussianCRFConv', 'PointConv', 'ResNetBBlock',
           'Upsampling', 'PointConvResNet', 'PointConvBig']
