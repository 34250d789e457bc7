"""Same export surface as the reference's models/__init__.py:1-2 for the dense path."""
from .common import MLP, Base, FastBatchNorm1d
from . import continuous_crf_conv, discrete_crf_conv, graph_ops, point_conv
from .discrete_crf_conv import DiscreteCRFConv
from .continuous_crf_conv import GuideGaussianCRFConv
from .continuous_crf_conv_big import ContinuousGaussianCRFConv
from .point_conv import (Baseline, BaselineDiscreteCRFSegNet, BaselineSegNet, DualCRFSegNet, CRFSegNet, CRFSegNet_Part, DepthwiseSeparablePointConv,
                         PointConvGassuianCRFNet, build_bipartite_graph, build_graph, knn_interpolate)
from .point_conv_big import PointConv, PointConvResNet, ResNetBBlock, Upsampling
from .point_conv_big import PointConvResNet as PointConvBig

__all__ = ['MLP', 'Base', 'FastBatchNorm1d', 'ContinuousGaussianCRFConv', 'GuideGaussianCRFConv',
           'DepthwiseSeparablePointConv', 'Baseline', 'PointConvGassuianCRFNet', 'CRFSegNet', 'CRFSegNet_Part',
           'BaselineSegNet', 'BaselineDiscreteCRFSegNet', 'DualCRFSegNet', 'DiscreteCRFConv', 'discrete_crf_conv', 'knn_interpolate', 'build_graph', 'build_bipartite_graph', 'continuous_crf_conv', 'point_conv',
           'graph_ops', 'PointConv', 'ResNetBBlock',
           'Upsampling', 'PointConvResNet', 'PointConvBig']
