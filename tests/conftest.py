import os
import sys

import pytest

# The suite checks the launches the modules issue one by one: the self-capturing training forward (crfconv_amd.train: autograph) stays
# off unless a test switches it on for itself (test_gpu_model.py: ..._captures_itself_...).
os.environ.setdefault('CRFCONV_AUTOGRAPH', '0')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
        return cache[name]
    return load


def sub(d, prefix):
    """Entries of a flat npz dict under 'prefix/' with the prefix stripped."""
    n = len(prefix) + 1
    return {k[n:]: v for k, v in d.items() if k.startswith(prefix + '/')}


@pytest.fixture(autouse=True)
def _tolerance_key(request):
    """Names the running test for gpu_util.assert_close (its bounds are tied to the errors recorded per call site)."""
    try:
        import gpu_util
    except Exception:                                      # CPU-only collection without torch extras: nothing to key
        yield
        return
    gpu_util.set_current_test(request.node.nodeid)
    yield


def pytest_sessionfinish(session, exitstatus):
    try:
        import gpu_util
        gpu_util.flush_recorded()
    except Exception:
        pass
