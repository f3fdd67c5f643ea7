"""Helpers shared by the parity tests: load the committed golden vectors
(produced from the real reference by ``tests/golden/_gen/make_golden.py``)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

with open(os.path.join(GOLDEN, 'MANIFEST.json')) as _f:
    MANIFEST = json.load(_f)

FULL = sorted(k for k, v in MANIFEST.items() if v['kind'] == 'full')
#: fixtures whose ORACLE takes minutes to build (SymPy differentiating a
#: 24-link pendulum); the oracle was checked against them once in the build
#: container (5e-16), the product is checked against them in every run.
HEAVY = {'config5_standin_24link_small'}
FULL_FAST = [k for k in FULL if k not in HEAVY]
SAMPLED = sorted(k for k, v in MANIFEST.items() if v['kind'] == 'sampled')


def load(name):
    return MANIFEST[name], np.load(os.path.join(GOLDEN, name + '.npz'))


def assert_close(actual, desired, rtol=1e-10, scale=None, what=''):
    """|a - d| <= rtol * max(|d|, scale): relative to the entry, with a floor
    ``scale`` at the magnitude of the terms that were summed (catastrophic
    cancellation cannot be held to 1e-10 of the *result*)."""
    actual = np.asarray(actual, dtype=float)
    desired = np.asarray(desired, dtype=float)
    assert actual.shape == desired.shape, (what, actual.shape, desired.shape)
    if scale is None:
        scale = float(np.max(np.abs(desired))) if desired.size else 1.0
    tol = rtol*np.maximum(np.abs(desired), scale)
    err = np.abs(actual - desired)
    bad = err > tol
    if bad.any():
        k = int(np.argmax(err/tol))
        raise AssertionError('%s: %d/%d entries off; worst at %d: %r vs %r'
                             % (what, bad.sum(), bad.size, k,
                                actual.flat[k], desired.flat[k]))
