"""crfconv_amd -- MI355X (gfx950) native hot path of continuous-CRF convolution.

Layout mirrors the reference's import surface for this path:
  crfconv_amd.models.PointConvBig / ContinuousGaussianCRFConv / PointConv / ResNetBBlock / MLP ...
  crfconv_amd.utils.nearest_neighbors.knn / knn_batch
  crfconv_amd.utils.cpp_subsampling.compute
All compute goes through libcrfconv_amd.so (include/crfconv_amd.h); there is no CPU fallback.
"""
import os as _os


def _pick_hw_queues():
    """ROCm deals HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin.  Under a process group THREE
    streams are busy at once when data.CollatePipeline overlaps preprocessing with training (training graphs, the collate
    graph, RCCL's own stream): with 4 queues the collective's stream shares a queue with the collate stream and waits
    behind a whole collate graph (7.1 ms per iteration instead of 5.8, measured under torch.distributed.run + RCCL); 8
    queues separate them.  WITHOUT a process group the default is the good mapping (8 queues measured 15 ms).  The
    variable is read when the HIP runtime initialises, so it is chosen here, at import, from the launcher's environment
    (RANK + WORLD_SIZE set = a rank of a process group); an explicit GPU_MAX_HW_QUEUES always wins."""
    if 'RANK' in _os.environ and 'WORLD_SIZE' in _os.environ:
        _os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


_pick_hw_queues()

from . import _lib, data, graph, models, ops, optim, train, utils          # noqa: E402
from .data import Data, MultiScaleData, multiscale_compute

__version__ = '0.1.0'
__all__ = ['models', 'utils', 'ops', 'optim', 'train', 'graph', 'data', 'Data', 'MultiScaleData', 'multiscale_compute']
