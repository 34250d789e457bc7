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


def _host_cpu_share():
    """CPUs this process may really run on: the cgroup quota when there is one, else the affinity mask."""
    n = len(_os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()                       # cgroup v2
    except (OSError, ValueError):
        try:
            quota = open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read().strip()               # cgroup v1
            period = open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read().strip()
        except OSError:
            return n
    if quota not in ('max', '-1'):
        n = min(n, max(1, int(quota) // int(period)))
    return n


def _fit_host_threads():
    """A container that shows all of the host's cores but grants a share of them (cgroup CPU quota; the MI355X boxes of this pool: 16 of
    128+) gets the WHOLE process throttled for the rest of a scheduler period once its threads have used the quota -- and torch's intra-op
    pool, sized from the core count, burns it by idle-spinning behind every small host op (the `torch.randperm` / `sort` of an eager
    collate, of a crop's shuffle).  Measured (`scratch/sync_probe.py`): one call in eight of an eager crop loop took 65-90 ms instead of
    1-6 ms; 16 crops of config 5 in 800 ms instead of 140.  At import the pool is cut to the quota when it is larger (never raised);
    CRFCONV_FIT_THREADS=0 leaves it alone."""
    if _os.environ.get('CRFCONV_FIT_THREADS', '1') == '0':
        return
    try:
        import torch
        share = _host_cpu_share()
        if torch.get_num_threads() > share:
            torch.set_num_threads(max(1, share))
    except Exception:        # noqa: BLE001  (never let a tuning step break the import)
        pass


_pick_hw_queues()

from . import _lib, data, graph, models, ops, optim, train, utils          # noqa: E402
from .data import Data, MultiScaleData, multiscale_compute

_fit_host_threads()

__version__ = '0.1.0'
__all__ = ['models', 'utils', 'ops', 'optim', 'train', 'graph', 'data', 'Data', 'MultiScaleData', 'multiscale_compute']
