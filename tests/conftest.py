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


# ---------------------------------------------------------------------------
# device backends.  "hip" = the product library on a real MI355X (marked gpu);
# "emu" = the same kernel sources compiled for the host by tests/emu (so the
# GPU-less container can still execute them).  The emulated build is test
# infrastructure and is injected explicitly here -- pyro2_amd never loads it
# on its own.
# ---------------------------------------------------------------------------
_CTX = {}


def _get_ctx(kind):
    from pyro2_amd import _lib, device
    if kind not in _CTX:
        if kind == "emu":
            sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
            import build_emu
            _lib.use_library(build_emu.build(), allow_backends=("host-emu",))
        else:
            _lib.use_library(None)
        _CTX[kind] = device.Context(0)
    return _CTX[kind]


@pytest.fixture(params=[pytest.param("emu"),
                        pytest.param("hip", marks=pytest.mark.gpu)])
def dev(request):
    """a pyro2_amd.device.Context on the requested backend"""
    ctx = _get_ctx(request.param)
    ctx.kind = request.param
    return ctx


@pytest.fixture
def hip():
    """real-GPU context (use together with @pytest.mark.gpu)"""
    ctx = _get_ctx("hip")
    ctx.kind = "hip"
    return ctx
