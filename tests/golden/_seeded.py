"""Seeded, platform-stable generators shared by make_golden.py (which runs the reference) and
the tests (which replay the same inputs through the oracle and the HIP path).

Everything is drawn from numpy's PCG64 (bit-stable across platforms and numpy versions), never
from torch's global RNG, so a fixture only has to store (seed, shapes) for large tensors.
"""
import zlib

import numpy as np
import torch


def _rng(seed, key=''):
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def fill_state_dict(shapes, seed):
    """shapes: {key: tuple}. Returns {key: torch tensor} with non-trivial, well-scaled values."""
    sd = {}
    for key in sorted(shapes):
        shape = tuple(int(s) for s in shapes[key])
        r = _rng(seed, key)
        if key.endswith('num_batches_tracked'):
            sd[key] = torch.zeros((), dtype=torch.long)
            continue
        if key.endswith('running_var'):
            v = r.uniform(0.5, 1.5, shape)
        elif key.endswith('running_mean'):
            v = r.uniform(-0.2, 0.2, shape)
        elif key.endswith('.c') or key == 'c':
            v = np.eye(shape[0]) + 0.1 * r.standard_normal(shape)
        elif 'batch_norm.weight' in key or (len(shape) == 1 and key.endswith('.weight')):
            v = r.uniform(0.5, 1.5, shape)
        elif len(shape) == 1:                       # biases
            v = r.uniform(-0.3, 0.3, shape)
        else:                                       # linear weights [out, in]
            v = r.uniform(-1.0, 1.0, shape) * (1.5 / np.sqrt(shape[1]))
        sd[key] = torch.from_numpy(np.asarray(v, dtype=np.float32))
    return sd


def make_cloud(seed, n, box=(1.0, 1.0, 1.0)):
    """n distinct float32 points, uniform in a box (tie-free with overwhelming probability)."""
    r = _rng(seed, 'cloud')
    return (r.random((n, 3)) * np.asarray(box)).astype(np.float32)


def uniform(seed, key, shape, lo=-1.0, hi=1.0):
    return _rng(seed, key).uniform(lo, hi, shape).astype(np.float32)


def integers(seed, key, shape, lo, hi):
    return _rng(seed, key).integers(lo, hi, shape)


def permutation(seed, key, n):
    return _rng(seed, key).permutation(n)


def projections(seed, key, numel, n=4):
    """n seeded +-1 probe vectors used to fingerprint tensors too large to store."""
    return _rng(seed, 'proj:' + key).integers(0, 2, (n, numel)).astype(np.float32) * 2 - 1


def build_multiscale(pos, knn_batch, ratios=(4, 4, 4, 4, 2), ks=(16, 16, 16, 16, 16), seed=0):
    """The reference collate (datasets/semantic3d_dataset.py:512-528) with a SEEDED subsample
    permutation in place of torch.randperm; `knn_batch(support, query, k) -> int64` is injected
    (the compiled reference when making goldens, the HIP kernel in tests)."""
    pos = np.asarray(pos, dtype=np.float32)
    ms = []
    for i, (ratio, k) in enumerate(zip(ratios, ks)):
        n = pos.shape[1]
        nbr = knn_batch(pos, pos, k)
        choice = permutation(seed, 'choice%d' % i, n)[: n // ratio]
        sub_pos = np.ascontiguousarray(pos[:, choice, :])
        sub_idx = np.ascontiguousarray(nbr[:, choice, :])
        up_idx = knn_batch(sub_pos, pos, 1)
        ms.append(dict(pos=pos, neighbor_idx=nbr, sub_idx=sub_idx, up_idx=up_idx))
        pos = sub_pos
    return ms
