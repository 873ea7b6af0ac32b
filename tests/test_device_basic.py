"""C-ABI surface, storage, transfers and ghost fill of libpyrohip."""
import ctypes
import os
import re

import numpy as np
import pytest

from pyro2_amd import _lib, build as hipbuild, device

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_binding_agree():
    """every function declared in include/pyrohip.h is bound, and vice versa"""
    hdr = open(os.path.join(ROOT, "include", "pyrohip.h")).read()
    declared = set(re.findall(r"\b(pyrohip_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)


def test_library_exports_every_symbol():
    """the gfx950 library loads on a GPU-less host and exports the whole ABI
    (no compute calls here)"""
    path = hipbuild.build()
    lib = ctypes.CDLL(path)
    for name in _lib.EXPORTS:
        assert hasattr(lib, name), name
    lib.pyrohip_backend.restype = ctypes.c_char_p
    assert lib.pyrohip_backend() == b"hip-gfx950"


def test_no_cpu_fallback(tmp_path):
    """the binding refuses a missing library and a non-HIP backend"""
    import subprocess
    import sys
    code = ("from pyro2_amd import _lib\n"
            "_lib.use_library(%r)\n"
            "try:\n    _lib.lib()\nexcept ImportError as e:\n    print('REFUSED', e)\n")
    out = subprocess.run([sys.executable, "-c", code % str(tmp_path / "nope.so")],
                         cwd=ROOT, capture_output=True, text=True).stdout
    assert "REFUSED" in out and "no CPU fallback" in out
    emu = os.path.join(ROOT, "tests", "_emu_build", "libpyrohip_emu.so")
    if os.path.exists(emu):
        out = subprocess.run([sys.executable, "-c", code % emu], cwd=ROOT,
                             capture_output=True, text=True).stdout
        assert "REFUSED" in out and "host-emu" in out


def test_roundtrip_and_rows(dev):
    rng = np.random.default_rng(0)
    s = device.DeviceState(dev, 13, 9, 4, [["outflow"] * 4] * 3)
    a = rng.standard_normal((21, 17, 3))
    s.upload(a)
    assert np.array_equal(s.download(), a)
    assert np.array_equal(s.download_var(1), a[:, :, 1])
    b = rng.standard_normal((5, 17, 3))
    s.upload_rows(7, b)
    a[7:12] = b
    assert np.array_equal(s.download(), a)
    assert np.array_equal(s.download_rows(3, 6), a[3:9])
    c = rng.standard_normal((21, 17))
    s.upload_var(2, c)
    a[:, :, 2] = c
    assert np.array_equal(s.download(), a)
    mn, mx = s.minmax(0)
    assert mn == a[4:-4, 4:-4, 0].min() and mx == a[4:-4, 4:-4, 0].max()
    mn, mx = s.minmax(2, buf=2)
    assert mn == a[2:-2, 2:-2, 2].min() and mx == a[2:-2, 2:-2, 2].max()


@pytest.mark.parametrize("ng", [4, 1])
@pytest.mark.parametrize("k", range(5))
def test_fill_bc_vs_reference(dev, golden, ng, k):
    """ArrayIndexer.fill_ghost golden vectors (array_indexer.py:150-274)"""
    g = golden("fill_bc")
    a = g[f"in_ng{ng}_{k}"]
    bcs = [str(b) for b in g[f"bc_ng{ng}_{k}"]]
    nx, ny = a.shape[0] - 2 * ng, a.shape[1] - 2 * ng
    s = device.DeviceState(dev, nx, ny, ng, [bcs])
    s.upload(a)
    s.fill_bc()
    assert np.array_equal(s.download()[:, :, 0], g[f"out_ng{ng}_{k}"])


def test_bad_arguments_raise(dev):
    with pytest.raises(_lib.PyroHipError):
        device.DeviceState(dev, 0, 4, 4, [["outflow"] * 4])
    s = device.DeviceState(dev, 8, 8, 2, [["outflow"] * 4])
    with pytest.raises(_lib.PyroHipError):   # advection needs ng >= 4
        s.adv_step(0, 0.1, 0.1, 1.0, 1.0, 0.01, 2)
