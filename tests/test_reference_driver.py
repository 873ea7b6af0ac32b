"""The drop-in boundary, proven with the reference's own driver: pyro/pyro_sim.py
of python-hydro/pyro2 is executed UNMODIFIED (runpy, from /root/reference) with
`pyro` resolving to this repository's alias package, so every `pyro.X` it imports
(RuntimeParameters, the solver modules, Simulation, CellCenterData2d, ...) is the
MI355X implementation -- and its results are compared with the reference's own
runs (tests/golden).  Needs the reference checkout: skipped on the GPU box, where
/root/reference does not exist."""
import os
import runpy
import sys

import numpy as np
import pytest

REF_DRIVER = "/root/reference/pyro/pyro_sim.py"
pytestmark = pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="no reference checkout here")


@pytest.fixture
def ref_pyro(dev, tmp_path, monkeypatch):
    """the class `Pyro` of the reference's pyro_sim.py on top of the alias package"""
    from pyro2_amd import device
    monkeypatch.setattr(device.Context, "_default", dev)
    monkeypatch.chdir(tmp_path)
    import matplotlib
    matplotlib.use("Agg")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    import pyro                          # noqa: F401  (this repository's alias package)
    assert os.path.realpath(pyro.__file__).startswith(os.path.realpath(root))
    ns = runpy.run_path(REF_DRIVER, run_name="reference_pyro_sim")
    src = open(REF_DRIVER).read()
    assert "pyro2_amd" not in src        # it really is the reference's file
    return ns["Pyro"]


def test_reference_driver_advection_smooth(ref_pyro, golden):
    """pyro_sim.py advection smooth inputs.smooth (BASELINE config 1 / pyro/test.py:93)"""
    g = golden("adv_smooth_0040")
    p = ref_pyro("advection")
    p.initialize_problem("smooth")
    ic = p.get_var("density")
    assert np.abs(np.asarray(ic) - g["ic"]).max() < 1e-15
    ic[:, :] = g["ic"]                   # the reference run's exp() bits
    del ic
    p.run_sim()
    assert p.sim.n == 40
    np.testing.assert_allclose(p.get_var("density").v(), g["gold"], rtol=1e-12, atol=0)
    if p.sim.cc_data.ctx.kind == "emu":   # bit-identical to the reference run on the same machine
        assert np.array_equal(p.get_var("density").v(), g["run"])


def test_reference_driver_compressible_sedov(ref_pyro, golden):
    """pyro_sim.py compressible sedov inputs.sedov at 64^2, 20 steps, against the
    reference's own run (comp_sedov_64_020.npz: dt sequence and end state)"""
    g = golden("comp_sedov_64_020")
    p = ref_pyro("compressible")
    p.initialize_problem("sedov", inputs_dict={"mesh.nx": 64, "mesh.ny": 64, "driver.max_steps": 20})
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    assert p.sim.n == 20
    assert np.abs(np.array(dts) - g["dts"][:20]).max() <= 1e-13 * np.abs(g["dts"]).max()
    fin = g["final"][4:-4, 4:-4]
    for n, name in enumerate(("density", "energy", "x-momentum", "y-momentum")):
        a = p.get_var(name).v()
        assert np.abs(a - fin[..., n]).max() <= 1e-12 * max(np.abs(fin[..., n]).max(), 1e-300), name
