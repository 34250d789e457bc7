"""torch.autograd bindings of the HIP kernels (through the C ABI; see include/crfconv_amd.h), one module per operator family:

    _base      conversions, per-stream zero words, the grid-barrier workspace and its failure check, run-time switches (`state`)
    defer      deferred parameter work of a backward pass: queues and the batched launches at its end
    crf        continuous CRF mean field (dense / wide), its matrices and their riders, the discrete CRF layer
    dense      per-point Linear, BatchNorm step counters, BatchNorm (+ LeakyReLU)
    mlp        Linear -> BatchNorm -> LeakyReLU blocks as single nodes (row-streaming, classifier head, coarse-level forms, groups)
    rows       residual join, LeakyReLU, row gather, neighbour max-pool
    loss       weighted soft-max cross-entropy
    pointconv  rel-pos moments, the PointConv node, the batched BatchNorm-1 prefold

Every op runs hand-written gfx950 kernels; torch supplies device memory, the stream and autograd.  The package namespace carries
every name of the modules (``ops.point_conv``, ``ops.crf_meanfield``, ... and the private ones the tests reach for); switches that
are flipped at run time live on ``ops.state``.
"""
from . import _base, defer, crf, dense, mlp, rows, loss, pointconv      # noqa: F401  (this order: see the modules' last lines)
from ..graph import NeighborTable      # noqa: F401

for _m in (_base, defer, crf, dense, mlp, rows, loss, pointconv):
    globals().update({_k: _v for _k, _v in vars(_m).items() if not _k.startswith('__')})
del _m

__all__ = ['linear', 'bn_act', 'crf_meanfield', 'gather_rows', 'neighbor_maxpool', 'relpos_moments', 'point_conv',
           'point_conv_prefold', 'cross_entropy', 'training_loss', 'NeighborTable']
