"""Drop-in for the reference's CPython extension ``grid_subsampling`` (imported as ``cpp_subsampling``,
utils/__init__.py:9; binding utils/cpp_wrappers/cpp_subsampling/wrapper.cpp:58-286):

    compute(points, features=None, classes=None, sampleDl=0.1, method='barycenters', verbose=0)

Same argument handling: everything after ``points`` is keyword-only (format "O|$OOfsi", wrapper.cpp:76),
points -> float32 [N, 3], features -> float32 [N, F], classes -> int32 [N] or [N, L]; returns a bare array
when only points are given, else a tuple (points, [features], [classes]) with classes always 2-D
(wrapper.cpp:269-276).  Errors are RuntimeError with the reference's messages.  `method` is validated and,
as in the reference (wrapper.cpp:72-91), otherwise ignored.  Rows come out in ascending voxel key."""
import numpy as np
import torch

from .. import _lib

_vp = _lib.ctypes.c_void_p


def _arr(obj, dtype, what):
    try:
        if torch.is_tensor(obj):
            obj = obj.detach().cpu().numpy()
        return np.ascontiguousarray(obj, dtype=dtype)
    except Exception:
        raise RuntimeError('Error converting input %s to numpy arrays of type %s'
                           % (what, 'int32' if dtype == np.int32 else 'float32'))


def compute(points, *, features=None, classes=None, sampleDl=0.1, method='barycenters', verbose=0):
    if method not in ('barycenters', 'voxelcenters'):
        raise RuntimeError('Error parsing method. Valid method names are "barycenters" and "voxelcenters" ')
    pts = _arr(points, np.float32, 'points')
    feats = None if features is None else _arr(features, np.float32, 'features')
    cls = None if classes is None else _arr(classes, np.int32, 'classes')
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise RuntimeError('Wrong dimensions : points.shape is not (N, 3)')
    if feats is not None and feats.ndim != 2:
        raise RuntimeError('Wrong dimensions : features.shape is not (N, d)')
    if cls is not None and cls.ndim > 2:
        raise RuntimeError('Wrong dimensions : classes.shape is not (N,) or (N, d)')
    N = pts.shape[0]
    fdim = feats.shape[1] if feats is not None else 0
    ldim = 1
    if cls is not None and cls.ndim == 2:
        ldim = cls.shape[1]
    if feats is not None and feats.shape[0] != N:
        raise RuntimeError('Wrong dimensions : features.shape is not (N, d)')
    if cls is not None and cls.shape[0] != N:
        raise RuntimeError('Wrong dimensions : classes.shape is not (N,) or (N, d)')
    if N < 1:
        raise RuntimeError('Error')
    if verbose > 0:
        print('Computing cloud pyramid with support points: ')
    out_p = np.empty((N, 3), dtype=np.float32)
    out_f = np.empty((N, fdim), dtype=np.float32) if feats is not None else None
    out_c = np.empty((N, ldim), dtype=np.int32) if cls is not None else None
    M = _lib.call('crfconv_grid_subsample', _vp(pts.ctypes.data), N,
                  _vp(feats.ctypes.data) if feats is not None else None, fdim,
                  _vp(cls.ctypes.data) if cls is not None else None, ldim if cls is not None else 0,
                  float(sampleDl), _vp(out_p.ctypes.data), _vp(out_f.ctypes.data) if out_f is not None else None,
                  _vp(out_c.ctypes.data) if out_c is not None else None, N)
    if M < 1:
        raise RuntimeError('Error')
    res = [out_p[:M].copy()]
    if out_f is not None:
        res.append(out_f[:M].copy())
    if out_c is not None:
        res.append(out_c[:M].copy())
    return res[0] if len(res) == 1 else tuple(res)
