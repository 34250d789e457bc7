"""crfconv_amd -- MI355X (gfx950) native hot path of continuous-CRF convolution.

Layout mirrors the reference's import surface for this path:
  crfconv_amd.models.PointConvBig / ContinuousGaussianCRFConv / PointConv / ResNetBBlock / MLP ...
  crfconv_amd.utils.nearest_neighbors.knn / knn_batch
  crfconv_amd.utils.cpp_subsampling.compute
All compute goes through libcrfconv_amd.so (include/crfconv_amd.h); there is no CPU fallback.
"""
from . import _lib, data, graph, models, ops, optim, utils
from .data import Data, MultiScaleData, multiscale_compute

__version__ = '0.1.0'
__all__ = ['models', 'utils', 'ops', 'optim', 'graph', 'data', 'Data', 'MultiScaleData', 'multiscale_compute']
