"""TEST INFRASTRUCTURE ONLY -- ctypes doors onto the CPU checkers.

  * ``liboracle.so``        our C restatements (oracle/knn_oracle.c, oracle/grid_oracle.c)
  * ``_ref/libref_knn.so``  the reference's own knn_.cxx + nanoflann, compiled unchanged
  * ``_ref/libref_grid.so`` the reference's own grid_subsampling.cpp + cloud.cpp, compiled unchanged

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_sz = ctypes.c_size_t


def build(ref=True):
    """Compile the checkers (idempotent). `ref` targets need /root/reference."""
    subprocess.check_call(['make', '-s', '-C', _HERE, 'all'])
    if ref and os.path.isdir(os.environ.get('CRFCONV_REFERENCE', '/root/reference')):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'ref'])


def _load(path):
    if not os.path.exists(path):
        raise FileNotFoundError(path + ' (run `make -C oracle` / `make -C oracle ref`)')
    return ctypes.CDLL(path)


_oracle = None
_refknn = None
_refgrid = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(os.path.join(_HERE, 'liboracle.so')):
            build(ref=False)
        _oracle = _load(os.path.join(_HERE, 'liboracle.so'))
    return _oracle


def have_ref():
    return (os.path.exists(os.path.join(_HERE, '_ref', 'libref_knn.so'))
            and os.path.exists(os.path.join(_HERE, '_ref', 'libref_grid.so')))


def ref_knn_lib():
    global _refknn
    if _refknn is None:
        _refknn = _load(os.path.join(_HERE, '_ref', 'libref_knn.so'))
    return _refknn


def ref_grid_lib():
    global _refgrid
    if _refgrid is None:
        _refgrid = _load(os.path.join(_HERE, '_ref', 'libref_grid.so'))
    return _refgrid


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def _knn_batch_call(fn, pts, queries, K):
    pts, queries = _f32(pts), _f32(queries)
    B, Np, dim = pts.shape
    Nq = queries.shape[1]
    out = np.zeros((B, Nq, K), dtype=np.int64)
    fn(_ptr(pts, _f32p), _sz(B), _sz(Np), _sz(dim), _ptr(queries, _f32p), _sz(Nq), _sz(K),
       _ptr(out, _i64p))
    return out


def oracle_knn_batch(pts, queries, K):
    """Brute-force restatement; pts [B,Np,3], queries [B,Nq,3] -> int64 [B,Nq,K]."""
    return _knn_batch_call(oracle_lib().oracle_knn_batch, pts, queries, K)


def oracle_knn(pts, queries, K):
    return oracle_knn_batch(np.asarray(pts)[None], np.asarray(queries)[None], K)[0]


def ref_knn_batch(pts, queries, K, omp=False):
    """The reference's cpp_knn_batch[_omp] (knn_.cxx:72-135), compiled unchanged."""
    lib = ref_knn_lib()
    return _knn_batch_call(lib.ref_knn_batch_omp if omp else lib.ref_knn_batch, pts, queries, K)


def ref_knn(pts, queries, K, omp=False):
    pts, queries = _f32(pts), _f32(queries)
    out = np.zeros((queries.shape[0], K), dtype=np.int64)
    lib = ref_knn_lib()
    fn = lib.ref_knn_omp if omp else lib.ref_knn
    fn(_ptr(pts, _f32p), _sz(pts.shape[0]), _sz(pts.shape[1]), _ptr(queries, _f32p),
       _sz(queries.shape[0]), _sz(K), _ptr(out, _i64p))
    return out


def knn_sq_dists(pts, queries, idx):
    """float32 squared distances (reference arithmetic) of given neighbours; single cloud."""
    pts, queries = _f32(pts), _f32(queries)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    out = np.zeros(idx.shape, dtype=np.float32)
    oracle_lib().oracle_knn_dists(_ptr(pts, _f32p), _sz(pts.shape[0]), _sz(pts.shape[1]),
                                  _ptr(queries, _f32p), _sz(queries.shape[0]), _ptr(idx, _i64p),
                                  _sz(idx.shape[1]), _ptr(out, _f32p))
    return out


def grid_keys(pts, dl):
    pts = _f32(pts)
    keys = np.zeros(pts.shape[0], dtype=np.uint64)
    oracle_lib().oracle_grid_keys(_ptr(pts, _f32p), ctypes.c_int64(pts.shape[0]), ctypes.c_float(dl),
                                  _ptr(keys, _u64p))
    return keys


def _grid_call(which, pts, feats, classes, dl):
    pts = _f32(pts)
    N = pts.shape[0]
    fdim = ldim = 0
    fp = cp = None
    of = oc = None
    if feats is not None:
        feats = _f32(feats)
        fdim = feats.shape[1]
        of = np.zeros((N, fdim), dtype=np.float32)
        fp = _ptr(feats, _f32p)
    if classes is not None:
        classes = np.ascontiguousarray(classes, dtype=np.int32)
        if classes.ndim == 1:
            classes = classes[:, None]
        ldim = classes.shape[1]
        oc = np.zeros((N, ldim), dtype=np.int32)
        cp = _ptr(classes, _i32p)
    op = np.zeros((N, 3), dtype=np.float32)
    ofp = _ptr(of, _f32p) if of is not None else None
    ocp = _ptr(oc, _i32p) if oc is not None else None
    if which == 'oracle':
        keys = np.zeros(N, dtype=np.uint64)
        fn = oracle_lib().oracle_grid_subsample
        fn.restype = ctypes.c_int64
        M = fn(_ptr(pts, _f32p), ctypes.c_int64(N), fp, ctypes.c_int(fdim), cp, ctypes.c_int(ldim),
               ctypes.c_float(dl), _ptr(op, _f32p), ofp, ocp, _ptr(keys, _u64p))
        keys = keys[:M]
    else:
        fn = ref_grid_lib().ref_grid_subsampling
        fn.restype = ctypes.c_long
        M = fn(_ptr(pts, _f32p), ctypes.c_long(N), fp, ctypes.c_int(fdim), cp, ctypes.c_int(ldim),
               ctypes.c_float(dl), _ptr(op, _f32p), ofp, ocp)
        keys = None
    return (op[:M], None if of is None else of[:M], None if oc is None else oc[:M], keys)


def oracle_grid_subsample(pts, feats=None, classes=None, dl=0.1):
    """-> (points[M,3], feats[M,F]|None, classes[M,L]|None, voxel_keys[M]) ascending by key."""
    return _grid_call('oracle', pts, feats, classes, dl)


def ref_grid_subsample(pts, feats=None, classes=None, dl=0.1):
    """The reference core (grid_subsampling.cpp:5-106); rows in its hash-map order."""
    return _grid_call('ref', pts, feats, classes, dl)[:3]
