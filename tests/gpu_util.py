"""Helpers shared by the -m gpu parity tests."""
import os

import numpy as np
import torch

DEV = 'cuda'


def t(a, dtype=None, dev=DEV):
    x = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        x = x.to(dtype)
    return x.to(dev)


def relerr(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def assert_close(a, b, tol, what=''):
    e = relerr(a, b)
    if os.environ.get('CRFCONV_TEST_REPORT'):             # measured error beside the bound (pytest -s): how much room a bound has
        print('[assert_close] %-44s err %.3e  tol %.1e' % (what, e, tol), flush=True)
    assert e <= tol, '%s: max err (rel. to max(1,|ref|)) %.3e > %.1e' % (what, e, tol)


def load_sd(module, sd_np):
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}
    module.load_state_dict(sd, strict=True)
    return module


def grads(module):
    return {k: p.grad for k, p in module.named_parameters() if p.grad is not None}


def assert_close_anchored(got, ref32, ref64, tol, what=''):
    """`got` may deviate from the float64 truth by `tol`, or by 4x the CPU float32 oracle's own
    rounding error where that is larger (ill-conditioned sums, e.g. gradients through BatchNorm)."""
    e, e32 = relerr(got, ref64), relerr(ref32, ref64)
    assert e <= max(tol, 4.0 * e32), '%s: err vs fp64 %.3e > max(%.1e, 4 x fp32-oracle err %.3e)' % (what, e, tol, e32)
