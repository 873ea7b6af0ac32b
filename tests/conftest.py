import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"),
                                  allow_pickle=False)
        return cache[name]
    return load


def max_rel_err(a, b):
    """scale-aware relative error with atol = 0 semantics:
    max |a-b| / max(|b|) over the array."""
    a = np.asarray(a)
    b = np.asarray(b)
    scale = np.abs(b).max()
    if scale == 0.0:
        return float(np.abs(a).max())
    return float(np.abs(a - b).max() / scale)
