"""The drop-in boundary, checked with the reference's OWN unit tests: the test files
of python-hydro/pyro2 (pyro/*/tests/test_*.py) are run unmodified by pytest with `pyro`
resolving to this repository's alias package (kernels on the HIP emulator).  Needs the
reference checkout: skipped on the GPU box.  The utilities those tests brought in
(integer-typed data, general restrict / prolong, pyro.multigrid.edge_coeffs) are also
pinned on the reference's results through tests/golden/mesh_utils.npz, which runs anywhere."""
import os
import subprocess
import sys

import numpy as np
import pytest

REF = "/root/reference/pyro"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (file, tests it holds).  Not here: advection_nonuniform (a solver outside SURVEY 8)
REF_TESTS = [("mesh/tests/test_patch.py", 25), ("mesh/tests/test_array_indexer.py", 4),
             ("mesh/tests/test_io.py", 2), ("util/tests/test_runparams.py", 6),
             ("multigrid/tests/test_multigrid_comps.py", 2), ("particles/tests/test_particles.py", 7),
             ("advection/tests/test_advection.py", 1), ("compressible/tests/test_compressible.py", 3),
             ("compressible/tests/test_eos.py", 1), ("compressible_rk/tests/test_compressible_rk.py", 1),
             ("diffusion/tests/test_diffusion.py", 1), ("swe/tests/test_swe.py", 3),
             ("tests/test_simulation.py", 4), ("tests/test_pyro.py", 2)]

PLUGIN = '''
import os, sys
ROOT = {root!r}
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import build_emu
from pyro2_amd import _lib, device
_lib.use_library(build_emu.LIB, allow_backends=("host-emu",))
device.Context._default = device.Context(0)
import matplotlib
matplotlib.use("Agg")
import pyro
assert os.path.realpath(pyro.__file__).startswith(os.path.realpath(ROOT))
'''


@pytest.mark.skipif(not os.path.exists(REF), reason="no reference checkout here")
def test_reference_unit_tests_pass_on_the_alias_package(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build()
    (tmp_path / "ref_unit_plugin.py").write_text(PLUGIN.format(root=ROOT))
    files = [os.path.join(REF, f) for f, _ in REF_TESTS]
    for f in files:
        assert "pyro2_amd" not in open(f).read()          # the reference's files, untouched
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), ROOT]), MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "ref_unit_plugin", "-p",
                        "no:cacheprovider", f"--rootdir={tmp_path}"] + files, cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert f"{sum(n for _, n in REF_TESTS)} passed" in r.stdout, tail


def test_restrict_prolong_general_grid(dev, golden, monkeypatch):
    """CellCenterData2d.restrict (N = 2, 4) / prolong on a rectangular grid with ng = 2
    (patch.py:640-736), bit for bit the reference's arrays"""
    from pyro2_amd import device
    from pyro2_amd.mesh import boundary as bnd
    from pyro2_amd.mesh import patch
    monkeypatch.setattr(device.Context, "_default", dev)
    z = golden("mesh_utils")
    g = patch.Grid2d(12, 8, ng=2, xmax=1.5, ymax=1.0)
    d = patch.CellCenterData2d(g)
    d.register_var("a", bnd.BC(xlb="outflow", xrb="outflow", ylb="periodic", yrb="periodic"))
    d.create()
    d.get_var("a")[:, :] = z["a"]
    assert np.array_equal(np.asarray(d.restrict("a")), z["r2"])
    assert np.array_equal(np.asarray(d.restrict("a", N=4)), z["r4"])
    assert np.array_equal(np.asarray(d.prolong("a")), z["p"])
    with pytest.raises(ValueError):
        d.restrict("a", N=3)


def test_integer_data(dev, monkeypatch):
    """dtype = int data (the reference's mesh tests use it): kept in that type on the host,
    ghost fill through the device"""
    from pyro2_amd import device
    from pyro2_amd.mesh import boundary as bnd
    from pyro2_amd.mesh import patch
    monkeypatch.setattr(device.Context, "_default", dev)
    g = patch.Grid2d(4, 6, ng=2)
    d = patch.CellCenterData2d(g, dtype=int)
    d.register_var("p", bnd.BC(xlb="periodic", xrb="periodic", ylb="reflect-even", yrb="outflow"))
    d.create()
    a = d.get_var("p")
    assert np.asarray(a).dtype == np.dtype(int)
    a.v()[:, :] = np.arange(24).reshape(4, 6) + 1
    d.fill_BC("p")
    a = np.asarray(d.get_var("p"))
    assert a.dtype == np.dtype(int)
    assert np.array_equal(a[0:2, 2:-2], a[4:6, 2:-2]) and np.array_equal(a[6:8, 2:-2], a[2:4, 2:-2])
    assert np.array_equal(a[2:-2, 1], a[2:-2, 2]) and np.array_equal(a[2:-2, 0], a[2:-2, 3])
    assert np.array_equal(a[2:-2, -1], a[2:-2, -3])


def test_edge_coeffs(golden):
    """pyro.multigrid.edge_coeffs.EdgeCoeffs and its restriction (edge_coeffs.py:1-54)"""
    from pyro2_amd.mesh import patch
    from pyro2_amd.multigrid import edge_coeffs
    z = golden("mesh_utils")
    g = patch.Grid2d(8, 12, ng=1, xmax=2.0, ymax=3.0)
    eta = g.scratch_array()
    eta[:, :] = z["eta"]
    e = edge_coeffs.EdgeCoeffs(g, eta)
    assert np.array_equal(np.asarray(e.x), z["ex"]) and np.array_equal(np.asarray(e.y), z["ey"])
    c = e.restrict()
    assert c.grid.nx == 4 and c.grid.ny == 6
    assert np.array_equal(np.asarray(c.x), z["cx"]) and np.array_equal(np.asarray(c.y), z["cy"])
