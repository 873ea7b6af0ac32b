"""The pyro-facing class surface (Pyro / Simulation / CellCenterData2d /
Grid2d / MG.CellCenterMG2d) driving the device path, against the reference's
golden files.  Runs on the emulated backend on CPU and on the GPU."""
import os

import numpy as np
import pytest

from conftest import max_rel_err


@pytest.fixture
def api(dev, tmp_path, monkeypatch):
    """make `dev` the default context of the host API; run in a scratch dir
    because Pyro writes inputs.auto into cwd (like the reference)"""
    from pyro2_amd import device
    monkeypatch.setattr(device.Context, "_default", dev)
    monkeypatch.chdir(tmp_path)
    return dev


def test_runparams_and_grid(tmp_path):
    from pyro2_amd.mesh import patch
    from pyro2_amd.util.runparams import RuntimeParameters
    f = tmp_path / "inputs"
    f.write_text("# c\n[driver]\ntmax = 2.5 ; end time\nname = abc\n[mesh]\nnx = 8\n")
    rp = RuntimeParameters()
    rp.load_params(str(f))
    assert rp.get_param("driver.tmax") == 2.5 and rp.get_param("mesh.nx") == 8
    assert rp.get_param("driver.name") == "abc"
    assert rp.param_comments["driver.tmax"] == "end time"
    with pytest.raises(KeyError):
        rp.get_param("nope.x")
    with pytest.raises(KeyError):
        rp.set_param("nope.x", 1)
    rp.set_param("new.key", 3, no_new=False)
    assert rp.get_param("new.key") == 3
    g = patch.Cartesian2d(4, 6, ng=2, xmax=2.0)
    assert (g.ilo, g.ihi, g.jlo, g.jhi, g.qx, g.qy) == (2, 5, 2, 7, 8, 10)
    assert g.dx == 0.5 and g.x[g.ilo] == 0.25 and g.x2d.shape == (8, 10)
    assert float(g.V[0, 0]) == g.dx * g.dy and g.Ax is g.Ly
    a = g.scratch_array()
    a[:, :] = np.arange(80).reshape(8, 10)
    assert a.v().shape == (4, 6) and a.ip(1)[0, 0] == a[3, 2] and a.jp(-1, buf=1)[0, 0] == a[1, 0]


def test_pyro_advection_smooth_regression(api, golden):
    """pyro/test.py:93 through the Pyro driver"""
    from pyro2_amd.pyro_sim import Pyro
    g = golden("adv_smooth_0040")
    p = Pyro("advection")
    p.initialize_problem("smooth")
    # same initial condition as the reference run (its exp() bits)
    ic = p.get_var("density")
    assert max_rel_err(ic, g["ic"]) < 1e-15
    ic[:, :] = g["ic"]
    p.run_sim()
    assert p.sim.n == 40
    np.testing.assert_allclose(p.get_var("density").v(), g["gold"], rtol=1e-12, atol=0)
    assert os.path.exists("inputs.auto")


def test_views_held_across_steps_stay_live(api):
    """reference scripts keep `dens = sim.cc_data.get_var("density")` across
    steps: the arrays are updated in place there, so the view shows every new
    state and writes through it count.  Here the state lives on the
    device; while such a view is alive the host copy is refreshed after every
    kernel and uploaded before the next one."""
    from pyro2_amd.pyro_sim import Pyro
    p = Pyro("advection")
    p.initialize_problem("smooth", inputs_dict={"mesh.nx": 16, "mesh.ny": 16})
    q = Pyro("advection")                       # twin without any view held
    q.initialize_problem("smooth", inputs_dict={"mesh.nx": 16, "mesh.ny": 16})
    dens = p.sim.cc_data.get_var("density")     # held across the steps below
    start = dens.v().copy()
    for _ in range(2):
        p.single_step()
        q.single_step()
    assert not np.array_equal(dens.v(), start)                       # the old view sees the new state
    assert np.array_equal(dens.v(), q.get_var("density").v())        # ... exactly
    dens.v()[:, :] = 7.0                                             # a write through the old view
    p.single_step()
    assert abs(float(np.mean(p.get_var("density").v())) - 7.0) < 1e-12
    # without a held view nothing is transferred: the deferred ghost fill stays deferred
    del dens
    p.single_step()
    p.sim.cc_data.fill_BC_all()
    assert p.sim.cc_data._fill_pending and not p.sim.cc_data._views_alive()
    a = p.get_var("density")                                         # any access carries the fill out
    assert not p.sim.cc_data._fill_pending
    g = p.sim.cc_data.grid
    assert np.array_equal(a[g.ilo - 1, g.jlo:g.jhi + 1], a[g.ihi, g.jlo:g.jhi + 1])   # periodic ghosts


def test_pyro_compressible_sedov(api, golden):
    from pyro2_amd.pyro_sim import Pyro
    g = golden("comp_sedov_64_020")
    nsteps = 20 if api.kind == "hip" else 5
    p = Pyro("compressible")
    p.initialize_problem("sedov", inputs_dict={"mesh.nx": 64, "mesh.ny": 64,
                                               "driver.max_steps": nsteps})
    assert np.array_equal(np.asarray(p.sim.cc_data.data), g["ic"])   # host init bit-identical
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    assert max_rel_err(np.array(dts), g["dts"][:nsteps]) < 1e-12
    if nsteps == 20:
        assert abs(p.sim.cc_data.t - float(g["t"])) < 1e-15
        U = np.asarray(p.sim.cc_data.data)
        assert max_rel_err(U[4:-4, 4:-4], g["final"][4:-4, 4:-4]) < 1e-12
    rho, u, v, pr = p.get_var("primitive")
    assert rho.v().min() > 0 and pr.v().min() > 0
    assert p.sim.cc_data.min("density") == rho.v().min()


def test_pyro_compressible_sod_ic_and_bcs(api, golden):
    """sod: initial condition identical to the reference's, reflecting y walls
    (default mesh BCs) resolved to even/odd per variable"""
    from pyro2_amd.pyro_sim import Pyro
    g = golden("comp_sod_x_0076")
    p = Pyro("compressible")
    p.initialize_problem("sod", inputs_file="inputs.sod.x",
                         inputs_dict={"driver.max_steps": 3})
    assert np.array_equal(np.asarray(p.sim.cc_data.data), g["ic"])
    bcs = p.sim.cc_data.BCs
    assert bcs["y-momentum"].ylb == "reflect-odd" and bcs["x-momentum"].ylb == "reflect-even"
    p.run_sim()
    assert max_rel_err(np.array([p.sim.dt]), g["dts"][2:3]) < 1e-12


@pytest.mark.parametrize("k,prob,d", [
    (0, "rt", {"mesh.nx": 16, "mesh.ny": 48}),
    (2, "hse", {"mesh.nx": 8, "mesh.ny": 32}),
])
def test_pyro_compressible_hse_boundaries(api, golden, k, prob, d):
    """rt / hse problems through Pyro: gravity + the `hse` user boundary
    (compressible/BC.py) on the device, against runs of the reference"""
    from pyro2_amd.pyro_sim import Pyro
    g = golden("comp_hse")
    pre = f"c{k}_"
    nsteps = len(g[pre + "dts"]) if api.kind == "hip" else 4
    p = Pyro("compressible")
    p.initialize_problem(prob, inputs_dict=dict(d, **{"driver.max_steps": nsteps}))
    ic = np.asarray(p.sim.cc_data.data)
    # cos / exp differ in the last bit between NumPy builds: not bit-identical
    assert np.array_equal(np.isnan(ic), np.isnan(g[pre + "ic"]))
    assert np.allclose(ic, g[pre + "ic"], rtol=4e-16, atol=1e-30, equal_nan=True)
    assert p.sim.cc_data.BCs["energy"].ylb == "hse"
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    assert max_rel_err(np.array(dts), g[pre + "dts"][:nsteps]) < 1e-12
    if nsteps == len(g[pre + "dts"]):
        U = np.asarray(p.sim.cc_data.data)
        # the x-momentum of the static atmosphere is pure round-off (1e-16)
        scale = np.maximum(np.abs(g[pre + "final"][4:-4, 4:-4]).max(axis=(0, 1)), 1e-3)
        assert (np.abs(U - g[pre + "final"])[4:-4, 4:-4] / scale).max() < 1e-12


def test_kh_ic(api, golden):
    from pyro2_amd.pyro_sim import Pyro
    g = golden("comp_stages")
    p = Pyro("compressible")
    p.initialize_problem("kh", inputs_dict={"mesh.nx": 16, "mesh.ny": 24, "driver.max_steps": 8})
    p.run_sim()
    p.sim.cc_data.fill_BC_all()
    U = np.asarray(p.sim.cc_data.data)
    assert max_rel_err(U, g["c6_U0"]) < 1e-12


def test_quad_ic(api, golden):
    from pyro2_amd.pyro_sim import Pyro
    g = golden("comp_quad_0606")
    p = Pyro("compressible")
    p.initialize_problem("quad", inputs_dict={"driver.max_steps": 0})
    assert np.array_equal(np.asarray(p.sim.cc_data.data), g["ic"])


def test_mg_class(api, golden):
    """mg_test_simple (pyro/multigrid/examples/mg_test_simple.py) through the
    CellCenterMG2d surface, 64^2 (256^2 on the GPU vs the stored golden)"""
    from pyro2_amd.multigrid import MG
    nx = 256 if api.kind == "hip" else 32
    a = MG.CellCenterMG2d(nx, nx, xl_BC_type="dirichlet", yl_BC_type="dirichlet",
                          xr_BC_type="dirichlet", yr_BC_type="dirichlet", verbose=0)
    a.init_zeros()
    rhs = -2.0 * ((1.0 - 6.0 * a.x2d**2) * a.y2d**2 * (1.0 - a.y2d**2) +
                  (1.0 - 6.0 * a.y2d**2) * a.x2d**2 * (1.0 - a.x2d**2))
    a.init_RHS(rhs)
    a.solve(rtol=1.e-11)
    v = a.get_solution()
    true = (a.x2d**2 - a.x2d**4) * (a.y2d**4 - a.y2d**2)
    e = v - true
    err = e.norm()
    if nx == 256:
        g = golden("mg_poisson_dirichlet_256")
        assert a.num_cycles == 7
        assert max_rel_err(v.v(), g["gold"]) < 1e-12
        assert abs(err - 1.60408e-06) < 1e-11
    else:
        assert abs(err - 1.02427e-04) < 1e-9        # mg_convergence.txt:4 (32^2)
    gx, gy = a.get_solution_gradient()
    assert gx.shape == v.shape
    assert a.residual_error < 1e-11 and a.grids[-1].get_var("r").shape == v.shape


def test_custom_problem_and_fill_bc(api):
    """add_problem + writes through get_var views reach the device"""
    from pyro2_amd.pyro_sim import Pyro

    def init(cc, rp):
        cc.get_var("density")[:, :] = 2.0

    p = Pyro("advection")
    p.add_problem("two", init, problem_params={"two.x": 1})
    p.initialize_problem("two", inputs_dict={"mesh.nx": 8, "mesh.ny": 8,
                                             "mesh.xlboundary": "outflow",
                                             "mesh.xrboundary": "outflow",
                                             "mesh.ylboundary": "periodic",
                                             "mesh.yrboundary": "periodic"})
    d = p.get_var("density")
    d.v()[:, :] = np.arange(64).reshape(8, 8)
    p.sim.cc_data.fill_BC_all()
    d = p.get_var("density")
    assert d[0, 4] == d[4, 4] and d[15, 7] == d[11, 7]        # outflow in x
    assert d[6, 0] == d[6, 8] and d[6, 15] == d[6, 7]          # periodic in y
    p.single_step()
    assert p.sim.n == 1 and p.sim.cc_data.t == p.sim.dt


def test_hdf5_roundtrip_and_benchmark_compare(api, tmp_path):
    """write() / io_pyro.read() / compare / PyroBenchmark: HDF5 through h5py when
    it is installed, util/h5pure.py (pure Python HDF5) otherwise"""
    from pyro2_amd.pyro_sim import PyroBenchmark
    from pyro2_amd.util import compare, io_pyro
    p = PyroBenchmark("advection", make_bench=True, bench_dir=str(tmp_path) + "/bench/")
    p.initialize_problem("smooth", inputs_dict={"driver.max_steps": 3})
    p.run_sim()
    s = io_pyro.read(str(tmp_path) + "/bench/smooth_0003")
    assert s.n == 3 and compare.compare(p.sim.cc_data, s.cc_data, rtol=0.0, atol=0.0) == 0
    q = PyroBenchmark("advection", comp_bench=True, bench_dir=str(tmp_path) + "/bench/")
    q.initialize_problem("smooth", inputs_dict={"driver.max_steps": 3})
    assert q.run_sim() == 0
    # a benchmark run takes the bit-faithful build unless told otherwise (the comparison is at
    # the reference's rtol = 1e-12, pyro_sim.py:353), and the file says which build wrote it
    assert p.rp.get_param("gpu.fast_math") == 0 and q.rp.get_param("gpu.fast_math") == 0
    assert int(s.restart_info["params"]["gpu.fast_math"]) == 0
    r = PyroBenchmark("advection", comp_bench=True, bench_dir=str(tmp_path) + "/bench/")
    r.initialize_problem("smooth", inputs_dict={"driver.max_steps": 3, "gpu.fast_math": 1})
    assert r.rp.get_param("gpu.fast_math") == 1


def test_restart_is_bit_identical(api, tmp_path):
    """row f3: write() -> Pyro.restart_problem() continues bit-identically
    (state interior, t, n and the dt history come from the file)"""
    from pyro2_amd.pyro_sim import Pyro
    d = {"mesh.nx": 32, "mesh.ny": 24, "sedov.r_init": 0.12}
    a = Pyro("compressible")
    a.initialize_problem("sedov", inputs_dict=dict(d, **{"driver.max_steps": 6}))
    a.run_sim()
    b = Pyro("compressible")
    b.initialize_problem("sedov", inputs_dict=dict(d, **{"driver.max_steps": 3}))
    b.run_sim()
    b.sim.write(str(tmp_path / "chk_0003"))
    c = Pyro("compressible")
    c.restart_problem(str(tmp_path / "chk_0003"), inputs_dict={"driver.max_steps": 6})
    assert c.sim.n == 3 and c.sim.cc_data.t == b.sim.cc_data.t
    assert c.rp.get_param("sedov.r_init") == 0.12 and c.sim.cc_data.grid.ny == 24
    c.run_sim()
    assert c.sim.n == 6 and c.sim.cc_data.t == a.sim.cc_data.t
    assert np.array_equal(np.asarray(c.sim.cc_data.data)[4:-4, 4:-4],
                          np.asarray(a.sim.cc_data.data)[4:-4, 4:-4])


def test_h5lite_tree(tmp_path):
    from pyro2_amd.util import h5lite
    with h5lite.NpzFile(str(tmp_path / "t.pyro.npz"), "w") as f:
        f.attrs["solver"] = "advection"
        g = f.create_group("state").create_group("density")
        g.create_dataset("data", data=np.arange(6.0).reshape(2, 3))
        g.attrs["xlb"] = "periodic"
        f.create_group("aux")
    with h5lite.NpzFile(str(tmp_path / "t.pyro.npz"), "r") as f:
        assert f.attrs["solver"] == "advection" and f.attrs.get("nope") is None
        assert list(f["state"]) == ["density"] and "aux" in f and "nope" not in f
        assert f["state"]["density"].attrs["xlb"] == "periodic"
        assert f["state/density/data"][1, 2] == 5.0
        assert list(f["aux"].attrs) == []


def test_pyro_import_alias():
    """code written against pyro2's module paths resolves to this package"""
    import pyro.multigrid.MG as MG
    from pyro.compressible import Simulation
    from pyro.mesh import patch
    from pyro.pyro_sim import Pyro
    import pyro2_amd.compressible
    import pyro2_amd.mesh.patch
    import pyro2_amd.multigrid.MG
    import pyro2_amd.pyro_sim
    assert Pyro is pyro2_amd.pyro_sim.Pyro
    assert patch is pyro2_amd.mesh.patch
    assert MG.CellCenterMG2d is pyro2_amd.multigrid.MG.CellCenterMG2d
    assert Simulation is pyro2_amd.compressible.Simulation
    with pytest.raises(ImportError):
        import pyro.no_such_solver  # noqa: F401


def test_pyro_ramp_problem(api, golden):
    """double Mach reflection through Pyro: IC and the time-dependent boundary"""
    from pyro2_amd.pyro_sim import Pyro
    g = golden("comp_ramp")
    nsteps = len(g["dts"]) if api.kind == "hip" else 3
    p = Pyro("compressible")
    p.initialize_problem("ramp", inputs_dict={"mesh.nx": 48, "mesh.ny": 12,
                                              "driver.max_steps": nsteps})
    assert np.array_equal(np.asarray(p.sim.cc_data.data), g["ic"])
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    assert max_rel_err(np.array(dts), g["dts"][:nsteps]) < 1e-12
    if nsteps == len(g["dts"]):
        scale = np.abs(g["final"]).max(axis=(0, 1))
        assert (np.abs(np.asarray(p.sim.cc_data.data) - g["final"]) / scale).max() < 1e-12


@pytest.mark.parametrize("k,prob,d", [
    (0, "heating", {"mesh.nx": 24, "mesh.ny": 24, "heating.r_src": 0.15, "heating.e_rate": 5.0}),
    (1, "plume", {"mesh.nx": 16, "mesh.ny": 32, "plume.r_pert": 0.6}),
    (2, "convection", {"mesh.nx": 16, "mesh.ny": 48, "convection.thickness": 0.5}),
])
def test_pyro_problem_sources(api, golden, k, prob, d):
    """heating / plume / convection through Pyro: IC, heating profile, run"""
    from pyro2_amd.pyro_sim import Pyro
    g = golden("comp_heating")
    pre = f"c{k}_"
    nsteps = len(g[pre + "dts"]) if api.kind == "hip" else 3
    p = Pyro("compressible")
    p.initialize_problem(prob, inputs_dict=dict(d, **{"driver.max_steps": nsteps}))
    ic = np.asarray(p.sim.cc_data.data)
    assert np.array_equal(np.isnan(ic), np.isnan(g[pre + "ic"]))
    # sqrt / exp / pow differ in the last bit between NumPy builds
    assert np.allclose(ic, g[pre + "ic"], rtol=2e-15, atol=1e-30, equal_nan=True)
    rate, prof = p.sim._heating()
    assert rate == float(g[pre + "e_rate"]) and np.allclose(prof, g[pre + "prof"], rtol=4e-16, atol=0)
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    assert max_rel_err(np.array(dts), g[pre + "dts"][:nsteps]) < 1e-12
    if nsteps == len(g[pre + "dts"]):
        fin = g[pre + "final"][4:-4, 4:-4]
        scale = np.maximum(np.abs(fin).max(axis=(0, 1)), 1e-3)
        assert (np.abs(np.asarray(p.sim.cc_data.data)[4:-4, 4:-4] - fin) / scale).max() < 1e-11


GENERAL_SOURCE_CASES = [
    ("sedov", {"mesh.nx": 24, "mesh.ny": 20, "driver.tmax": 100.0}),
    ("plume", {"mesh.nx": 16, "mesh.ny": 32, "plume.r_pert": 0.6}),
    ("convection", {"mesh.nx": 16, "mesh.ny": 48, "convection.thickness": 0.5}),
    ("rt", {"mesh.nx": 16, "mesh.ny": 48, "mesh.ylboundary": "reflect", "mesh.yrboundary": "reflect",
            "mesh.xlboundary": "reflect", "mesh.xrboundary": "outflow"}),
]


@pytest.mark.parametrize("k", range(4))
@pytest.mark.parametrize("fast", [0, 1])
def test_arbitrary_problem_source_vs_reference(api, golden, k, fast):
    """an arbitrary source_terms() callback (tests/general_source.py: four components,
    state dependent) evaluated on the host twice per step, applied on the device where
    compressible/simulation.py applies it (interface states, predictor, corrector):
    10 steps of the REFERENCE run with the same callback (oracle/gen_golden.py
    comp_general_source) -- outflow; hse + gravity; ambient + sponge + gravity;
    reflecting walls + gravity"""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from general_source import source_terms
    from pyro2_amd.pyro_sim import Pyro
    g = golden("comp_general_source")
    prob, d = GENERAL_SOURCE_CASES[k]
    pre = f"c{k}_"
    p = Pyro("compressible")
    p.initialize_problem(prob, inputs_dict=dict(d, **{"gpu.fast_math": fast}))
    p.sim.problem_source, p.sim.problem_heating = source_terms, None
    p.sim._heat_cache = None
    assert p.sim._host_source() and not p.sim.can_evolve_many()
    ic = np.asarray(p.sim.cc_data.data)
    assert np.allclose(ic, g[pre + "ic"], rtol=2e-15, atol=1e-30, equal_nan=True)
    dts = []
    for _ in range(len(g[pre + "dts"])):
        p.single_step()
        dts.append(p.sim.dt)
    tol = 1e-9 if fast else 1e-12
    assert max_rel_err(np.array(dts), g[pre + "dts"]) < tol
    fin = g[pre + "final"][4:-4, 4:-4]
    scale = np.maximum(np.abs(fin).max(axis=(0, 1)), 1e-3)
    err = (np.abs(np.asarray(p.sim.cc_data.data)[4:-4, 4:-4] - fin) / scale).max()
    assert err < (1e-8 if fast else 1e-11), err


@pytest.mark.parametrize("prob,d", [("heating", {"mesh.nx": 24, "mesh.ny": 24, "heating.r_src": 0.15, "heating.e_rate": 5.0}),
          ("plume", {"mesh.nx": 16, "mesh.ny": 32, "plume.r_pert": 0.6}),
          ("convection", {"mesh.nx": 16, "mesh.ny": 48, "convection.thickness": 0.5})])
def test_host_source_equals_device_heating(api, prob, d):
    """the three heating problems through the host-evaluated source path
    (gpu.host_source = 1, their own source_terms()) give bit for bit what the device
    heating profile gives with the staged kernels: same arithmetic, other plumbing"""
    from pyro2_amd.pyro_sim import Pyro
    res = []
    for host in (0, 1):
        p = Pyro("compressible")
        p.initialize_problem(prob, inputs_dict=dict(d, **{"gpu.host_source": host,
                                                          "gpu.kernel_set": 0}))
        assert p.sim._host_source() == bool(host)
        for _ in range(6):
            p.single_step()
        res.append((p.sim.dt, np.array(p.sim.cc_data.data)[4:-4, 4:-4]))
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("prob,d", [
    ("acoustic_pulse", {"mesh.nx": 24, "mesh.ny": 24}),
    ("advect", {"mesh.nx": 16, "mesh.ny": 20}),
    ("bubble", {"mesh.nx": 32, "mesh.ny": 64}),
    ("gresho", {"mesh.nx": 20, "mesh.ny": 20}),
    ("rt2", {"mesh.nx": 24, "mesh.ny": 48}),
    ("rt_multimode", {"mesh.nx": 24, "mesh.ny": 48}),
])
def test_problem_initial_conditions(api, golden, prob, d):
    """the remaining compressible problem set-ups produce the reference's initial
    state (to the last bits of exp / cos / log between NumPy builds) and boundaries"""
    from pyro2_amd.pyro_sim import Pyro
    g = golden("comp_problem_ics")
    p = Pyro("compressible")
    p.initialize_problem(prob, inputs_dict=dict(d, **{"driver.max_steps": 0}))
    ic = np.asarray(p.sim.cc_data.data)
    assert np.array_equal(np.isnan(ic), np.isnan(g[prob]))
    assert np.allclose(ic, g[prob], rtol=4e-15, atol=1e-18, equal_nan=True)
    bc = p.sim.cc_data.BCs["density"]
    assert [bc.xlb, bc.xrb, bc.ylb, bc.yrb] == [str(b).replace("reflect", "reflect-even")
                                                  if str(b) == "reflect" else str(b)
                                                  for b in g[prob + "_bc"]]
    if api.kind == "hip" and prob != "bubble":   # bubble is under-resolved at this size
        p.sim.max_steps = 5                          # (the reference fails on it too)
        p.run_sim()
        assert p.sim.n == 5 and np.isfinite(np.asarray(p.sim.cc_data.data)[4:-4, 4:-4]).all()


@pytest.mark.parametrize("k", range(2))
def test_pyro_compressible_spherical(api, golden, k):
    """mesh.grid_type = SphericalPolar through Pyro: the grid's geometry arrays,
    the problem set-up and a short run against the reference (inputs.sedov.
    spherical, inputs.advect.spherical.64 at reduced sizes)"""
    from pyro2_amd.mesh import patch
    from pyro2_amd.pyro_sim import Pyro
    g = golden("comp_spherical")
    pre = f"c{k}_"
    meta = g[pre + "meta"]
    nx, ny, ng = int(meta[0]), int(meta[1]), int(meta[2])
    prob = str(g[pre + "problem"])
    inp = "inputs.sedov.spherical" if prob == "sedov" else "inputs.advect.spherical.64"
    nsteps = len(g[pre + "dts"])
    p = Pyro("compressible")
    p.initialize_problem(prob, inputs_file=inp, inputs_dict={"mesh.nx": nx, "mesh.ny": ny,
                                                             "driver.max_steps": nsteps})
    grid = p.sim.cc_data.grid
    assert isinstance(grid, patch.SphericalPolar) and grid.coord_type == 1
    geo = grid.device_geometry()
    for n in ("Lx", "Ly", "Ax", "Ay", "V", "dlogAx", "dlogAy", "x2d", "sint", "sinb", "sinc"):
        assert np.allclose(geo[n], g[pre + "g_" + n], rtol=4e-15, atol=0.0), n
    assert np.allclose(np.asarray(p.sim.cc_data.data), g[pre + "ic"], rtol=4e-15, atol=0.0)
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    assert np.abs(np.array(dts) / g[pre + "dts"] - 1).max() < 1e-11
    got = np.asarray(p.sim.cc_data.data)
    I = (slice(ng, -ng), slice(ng, -ng))
    scale = np.maximum(np.abs(g[pre + "after"][I]).max(axis=(0, 1)), 1e-3)
    assert (np.abs(got - g[pre + "after"])[I] / scale).max() < 1e-10
    # HLLC is refused on this grid like in the reference (compressible/simulation.py:206-208)
    q = Pyro("compressible")
    with pytest.raises(BaseException):
        q.initialize_problem(prob, inputs_file=inp, inputs_dict={"mesh.nx": nx, "mesh.ny": ny,
                                                                 "compressible.riemann": "HLLC"})


def test_burgers_reference_regression(api, golden):
    """pyro/test.py:99 -- burgers test inputs.test against burgers/tests/test_0051.h5 (128^2,
    51 steps, tracer particles on); on the emulator the first steps only"""
    from pyro2_amd.pyro_sim import Pyro
    g = golden("burgers_test_0051")
    p = Pyro("burgers")
    p.initialize_problem("test", inputs_file="inputs.test")
    ic = np.asarray(p.sim.cc_data.data)
    assert np.allclose(ic, g["ic"], rtol=2e-15, atol=1e-30)
    nsteps = int(g["nsteps"]) if api.kind == "hip" else 3
    dts = []
    while not p.sim.finished() and len(dts) < nsteps:
        p.single_step()
        dts.append(p.sim.dt)
    assert max_rel_err(np.array(dts), g["dts"][:nsteps]) < 1e-12
    if api.kind == "hip":
        assert p.sim.finished() and p.sim.n == 51
        U = np.stack([p.get_var(nm).v() for nm in ("x-velocity", "y-velocity")], axis=-1)
        assert np.abs(U - g["gold"]).max() < 1e-11
        assert np.abs(U - g["run"]).max() < 1e-12


@pytest.mark.gpu
def test_compressible_rk_reference_regression(hip, golden, tmp_path, monkeypatch):
    """pyro/test.py:103 -- compressible_rk rt inputs.rt against compressible_rk/tests/rt_1835.h5
    (64 x 192, RK2 method of lines, gravity, 1835 steps to t = 3)"""
    from pyro2_amd import device
    from pyro2_amd.pyro_sim import Pyro
    monkeypatch.setattr(device.Context, "_default", hip)
    monkeypatch.chdir(tmp_path)
    g = golden("comp_rk_rt_1835")
    p = Pyro("compressible_rk")
    p.initialize_problem("rt", inputs_file="inputs.rt")
    p.run_sim()
    assert p.sim.cc_data.t == float(g["time"])
    assert p.sim.n == int(g["nsteps"]) == 1835
    U = np.stack([p.get_var(nm).v() for nm in ("density", "energy", "x-momentum", "y-momentum")],
                 axis=-1)
    # 1835 steps of a Rayleigh-Taylor instability: compare like the reference's own
    # regression tool does (pyro/util/compare.py: relative to the field's magnitude)
    scale = np.abs(g["gold"]).max(axis=(0, 1))
    assert (np.abs(U - g["gold"]) / scale).max() < 1e-9


@pytest.mark.parametrize("problem,extra", [
    ("smooth", {}),                                                   # periodic: several steps per launch
    ("tophat", {"mesh.xlboundary": "outflow", "mesh.xrboundary": "outflow"}),   # single steps inside the call
])
def test_advection_run_sim_batches_steps(api, problem, extra):
    """Pyro.run_sim hands batches of steps to pyrohip_adv_evolve when nothing happens between
    them (verbose = 0, no output, no plot): same data (ghost cells included), step count, time,
    dt and dt_old as the step-by-step loop of pyro_sim.py:241-281 -- the first steps grow by
    max_dt_change, the last one lands on tmax (simulation_null.py:222-244)"""
    from pyro2_amd.pyro_sim import Pyro
    got = {}
    for batch in (False, True):
        p = Pyro("advection")
        p.initialize_problem(problem, inputs_dict=dict({"mesh.nx": 32, "mesh.ny": 48, "driver.tmax": 0.3,
                                                        "gpu.fast_math": 0, "gpu.adv_steps_per_launch": 3},
                                                       **extra))
        assert p.sim.can_evolve_many()
        if not batch:
            p.sim.can_evolve_many = lambda: False
        p.run_sim()
        got[batch] = (np.array(p.sim.cc_data.data), p.sim.n, p.sim.cc_data.t, p.sim.dt, p.sim.dt_old)
        parts = p.sim.particles       # inputs.smooth carries tracer particles (constant velocity)
        got[batch] += (None if parts is None else np.array(parts.get_positions()),)
    assert got[True][1:5] == got[False][1:5] and got[True][1] > 10
    assert np.array_equal(got[True][0], got[False][0])
    assert (got[True][5] is None) == (got[False][5] is None)
    assert got[True][5] is None or np.array_equal(got[True][5], got[False][5])


def test_spherical_host_side_boundary_is_read(api):
    """a SphericalPolar run with a boundary type that only has a HOST
    callback (define_bc without a device code) -- the one-launch spherical kernel reads ghost
    cells through the outflow / reflect / periodic index maps and must not be taken when the
    ghost cells hold what a callback wrote.  Default kernel set against the staged set."""
    from pyro2_amd.mesh import boundary as bnd
    from pyro2_amd.pyro_sim import Pyro

    def inflow(bc_name, bc_edge, variable, ccdata, ivars=None):
        g = ccdata.grid
        assert bc_edge == "xrb"
        a = ccdata.data[:, :, ccdata.names.index(variable)]
        a[g.ihi + 1:, :] = {"density": 1.5, "energy": 3.0, "x-momentum": -0.4, "y-momentum": 0.0}[variable]

    bnd.define_bc("inflow_test", inflow, is_solid=False)
    try:
        out = []
        for kset in (-1, 0):
            p = Pyro("compressible")
            p.initialize_problem("sedov", inputs_file="inputs.sedov.spherical",
                                 inputs_dict={"mesh.nx": 64, "mesh.ny": 32, "driver.max_steps": 6,
                                              "mesh.xrboundary": "inflow_test", "gpu.kernel_set": kset,
                                              "gpu.fast_math": 0})
            assert any(p.sim.cc_data._has_host_bc(n) for n in p.sim.cc_data.names)
            p.run_sim()
            assert p.sim.n == 6
            out.append(np.asarray(p.sim.cc_data.data).copy())
        g = p.sim.cc_data.grid
        I = (slice(g.ng, -g.ng), slice(g.ng, -g.ng))
        assert np.array_equal(out[0][I], out[1][I])
        # the inflow is felt: the last interior column is no longer the ambient gas
        assert np.abs(out[0][g.ihi, g.ng:-g.ng, 0] - 1.0).max() > 1e-5
    finally:
        for d in (bnd.ext_bcs, bnd.bc_solid):
            d.pop("inflow_test", None)


def test_advection_zero_time_step_ends(api):
    """driver.cfl = 0 gives dt = 0; the batched path hands nothing to the device
    and the driver must fall back to single steps (max_steps zero-length steps, like the
    reference) instead of spinning"""
    from pyro2_amd.pyro_sim import Pyro
    p = Pyro("advection")
    p.initialize_problem("smooth", inputs_file="inputs.smooth",
                         inputs_dict={"mesh.nx": 16, "mesh.ny": 16, "driver.cfl": 0.0, "driver.max_steps": 3,
                                      "particles.do_particles": 0})
    before = np.asarray(p.sim.cc_data.data).copy()
    p.run_sim()
    assert p.sim.n == 3 and p.sim.cc_data.t == 0.0
    g = p.sim.cc_data.grid
    assert np.array_equal(np.asarray(p.sim.cc_data.data)[g.ng:-g.ng, g.ng:-g.ng], before[g.ng:-g.ng, g.ng:-g.ng])


def test_benchmark_run_honours_the_inputs_file_build(api, tmp_path):
    """PyroBenchmark defaults to the bit-faithful build, but a gpu.fast_math the
    inputs FILE names wins (the default has the strength of a default)"""
    from pyro2_amd.pyro_sim import PyroBenchmark
    src = open(os.path.join(os.path.dirname(__import__("pyro2_amd").__file__), "advection", "problems",
                            "inputs.smooth")).read()
    f = tmp_path / "inputs.mine"
    f.write_text(src + "\n[gpu]\nfast_math = 1\n")
    p = PyroBenchmark("advection")
    p.initialize_problem("smooth", inputs_file=str(f), inputs_dict={"driver.max_steps": 1})
    assert int(p.rp.get_param("gpu.fast_math")) == 1
    q = PyroBenchmark("advection")
    q.initialize_problem("smooth", inputs_file="inputs.smooth", inputs_dict={"driver.max_steps": 1})
    assert int(q.rp.get_param("gpu.fast_math")) == 0
