"""Drop-in for the reference's Cython module ``nearest_neighbors`` (utils/nearest_neighbors/knn.pyx:33-109):
``knn(pts, queries, K, omp=False)`` and ``knn_batch(pts, queries, K, omp=False)`` take numpy arrays or
CPU tensors, coerce to C-contiguous float32 (knn.pyx:55-56, 95-96) and return a fresh ``np.int64`` array.
The search itself is the HIP grid kNN in libcrfconv_amd.so (`omp` is accepted and ignored: the GPU
kernel is already parallel over queries and clouds).  ``knn_batch_device`` is the zero-copy form
for tensors that already live on the GPU."""
import numpy as np
import torch

from .. import _lib
from ..graph import ptr, stream_ptr

_vp = _lib.ctypes.c_void_p


def _np_f32(a):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def knn(pts, queries, K, omp=False):
    pts_c, queries_c = _np_f32(pts), _np_f32(queries)
    if pts_c.ndim != 2 or queries_c.ndim != 2:
        raise ValueError('knn expects [Np, dim] and [Nq, dim] arrays')
    indices = np.zeros((queries_c.shape[0], K), dtype=np.int64)
    fn = 'crfconv_knn_omp' if omp else 'crfconv_knn'
    _lib.call(fn, _vp(pts_c.ctypes.data), pts_c.shape[0], pts_c.shape[1], _vp(queries_c.ctypes.data),
              queries_c.shape[0], int(K), _vp(indices.ctypes.data))
    return indices


def knn_batch(pts, queries, K, omp=False):
    pts_c, queries_c = _np_f32(pts), _np_f32(queries)
    if pts_c.ndim != 3 or queries_c.ndim != 3:
        raise ValueError('knn_batch expects [B, Np, dim] and [B, Nq, dim] arrays')
    indices = np.zeros((pts_c.shape[0], queries_c.shape[1], K), dtype=np.int64)
    fn = 'crfconv_knn_batch_omp' if omp else 'crfconv_knn_batch'
    _lib.call(fn, _vp(pts_c.ctypes.data), pts_c.shape[0], pts_c.shape[1], pts_c.shape[2],
              _vp(queries_c.ctypes.data), queries_c.shape[1], int(K), _vp(indices.ctypes.data))
    return indices


def knn_batch_device(pts, queries, K, out_dtype=torch.int64):
    """pts [B, Np, 3], queries [B, Nq, 3] float32 CUDA tensors -> [B, Nq, K] (int64 or int32) on the
    same device, enqueued on the current stream (no host sync)."""
    if not (pts.is_cuda and queries.is_cuda):
        raise _lib.CrfConvError('knn_batch_device needs CUDA tensors (use knn_batch for host arrays)')
    pts = pts.detach().to(torch.float32).contiguous()
    queries = queries.detach().to(torch.float32).contiguous()
    B, Np, dim = pts.shape
    Nq = queries.shape[1]
    out = torch.empty((B, Nq, K), dtype=out_dtype, device=pts.device)
    nbytes = _lib.load().crfconv_knn_batch_dev_workspace(B, Np, Nq, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    o64 = ptr(out) if out_dtype == torch.int64 else None
    o32 = ptr(out) if out_dtype == torch.int32 else None
    if o64 is None and o32 is None:
        raise ValueError('out_dtype must be torch.int64 or torch.int32')
    _lib.call('crfconv_knn_batch_dev', ptr(pts), B, Np, dim, ptr(queries), Nq, int(K), o64, o32, ptr(ws), nbytes,
              stream_ptr())
    return out
