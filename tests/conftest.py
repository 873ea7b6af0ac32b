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
    _parallel_cpu_suite(config)


def _parallel_cpu_suite(config):
    """The CPU suite (-m "not gpu") executes the HIP kernels on a single-threaded
    emulator; spread it over a few pytest-xdist workers so that it stays within
    a few minutes.  The GPU suite is never parallelised (one context, one GPU).
    PYRO_TEST_WORKERS=0 disables, an explicit -n wins."""
    if hasattr(config, "workerinput"):          # we are a worker already
        return
    if (config.getoption("markexpr", "") or "").replace(" ", "") != "notgpu":
        return
    # (default: the cores of the box less two, between 2 and 8 workers)
    n = int(os.environ.get("PYRO_TEST_WORKERS", str(min(8, max(2, (os.cpu_count() or 4) - 2)))))
    if n <= 1 or not config.pluginmanager.hasplugin("xdist"):
        return
    if getattr(config.option, "numprocesses", None) or getattr(config.option, "dist", "no") != "no":
        return
    # build the shared test libraries once, before the workers race for them
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build()
    from oracle import orc
    orc.build()
    config.option.numprocesses = n
    config.option.tx = ["popen"] * n
    config.option.dist = "load"


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


def elementwise_err(a, b, floor):
    """max over the ELEMENTS of |a - b| / (|b| + floor): an element-wise rtol
    with atol = rtol * floor, floor = the local scale of that variable (e.g. its
    ambient value), so that cells eight decades below the array maximum are
    held to the same relative tolerance as the peak (the array-wide
    max_rel_err above lets an O(1e-2) relative error in an ambient cell pass)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) / (np.abs(b) + floor)).max())


def comp_floors(ref, gamma=1.4):
    """per-variable local scales of a compressible state (density, energy,
    x-momentum, y-momentum): the smallest density / energy magnitudes of the
    reference (the ambient gas) and, for the momenta, rho * c of that gas"""
    r = np.asarray(ref)
    rho = float(np.abs(r[..., 0]).min())
    ener = float(np.abs(r[..., 1]).min())
    # ambient sound speed from the smallest internal energy: p = (gamma - 1) rho e
    c = float(np.sqrt(gamma * (gamma - 1.0) * ener / max(rho, 1e-300)))
    return np.array([rho, ener, rho * c, rho * c])


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
