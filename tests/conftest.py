import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class Golden:
    """Read-only view of one tests/golden/*.npz with '/'-separated keys."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)

    def __getitem__(self, key):
        return self._z[key.replace('/', '__')]

    def __contains__(self, key):
        return key.replace('/', '__') in self._z.files

    def keys(self):
        return [k.replace('__', '/') for k in self._z.files]


_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture(scope='session')
def gold():
    return golden


def relerr(a, b):
    """max |a-b| / max(|b|, tiny), NaN-aware: NaNs must coincide."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), 'NaN pattern differs'
    inf = np.isinf(b)
    assert np.array_equal(a[inf], b[inf]), 'infinities differ'
    ok = ~(nan_b | inf)
    if not ok.any():
        return 0.
    denom = np.maximum(np.abs(b[ok]), 1e-300)
    return float(np.max(np.abs(a[ok] - b[ok]) / denom))
