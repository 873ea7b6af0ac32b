"""The x-slab decomposition BEHIND pyro's class surface: N gloo processes each
run the SAME script -- `Pyro("compressible")` / `Pyro("advection")`, `initialize_problem`,
`run_sim()` or `single_step()` -- and together step ONE problem: grid_setup hands every process its
slab's Grid2d, CellCenterData2d.fill_BC_all exchanges halo rows, compute_timestep takes the global
CFL minimum, write() gathers.  Everything must be BIT-IDENTICAL to the single-domain `Pyro` run of
the same library, dt sequence included (reference: pyro/pyro_sim.py:182-189,241-281,
pyro/simulation_null.py:10-69,222-244).  Emulated backend; halo rows travel over gloo
(tests/host_comm.py) -- directly in the stepped path, through the library's communicator hooks
(tests/emu/comm_emu.cpp) where the steps are batched on the device (Simulation.evolve_many)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DRIVER = "/root/reference/pyro/pyro_sim.py"


def blast_problem(cc, rp):
    """a hot spot OFF the centre, written like a user's init_data: from the grid's coordinate
    arrays, untouched by any knowledge of slabs"""
    g = cc.grid
    gamma = rp.get_param("eos.gamma")
    dens, ener = cc.get_var("density"), cc.get_var("energy")
    dens[:, :] = 1.0
    cc.get_var("x-momentum")[:, :] = 0.0
    cc.get_var("y-momentum")[:, :] = 0.0
    r = np.sqrt((np.asarray(g.x2d) - 0.21)**2 + (np.asarray(g.y2d) - 0.55)**2)
    p = np.where(r < 0.09, 40.0, 1.e-3)
    ener[:, :] = p / (gamma - 1.0)


COMP = {"mesh.nx": 64, "mesh.ny": 24, "mesh.xlboundary": "outflow", "mesh.xrboundary": "outflow",
        "mesh.ylboundary": "outflow", "mesh.yrboundary": "outflow", "driver.max_steps": 7,
        "driver.tmax": 1.0, "gpu.fast_math": 0}


def run_case(case, tmp_out=None):
    """the script every process runs (also the single-domain comparison: same function, no
    decomposition installed).  Returns what is compared."""
    from pyro2_amd.pyro_sim import Pyro
    res = {}
    if case in ("comp_steps", "comp_batched", "comp_write"):
        p = Pyro("compressible")
        p.add_problem("blast", blast_problem)
        over = dict(COMP)
        if case == "comp_steps":      # every other rank marches rows, the rest use the tile kernel
            from pyro2_amd import decomp
            dec = decomp.active_decomposition()
            over["gpu.kernel_set"] = 1 + (dec.rank % 2 if dec else 0)
        p.initialize_problem("blast", inputs_dict=over)
        dts = []
        if case == "comp_steps":
            while not p.sim.finished():
                p.single_step()
                dts.append(p.sim.dt)
        else:
            keep = p.sim.evolve_many

            def spy(n):
                out = keep(n)
                dts.extend(float(x) for x in out)
                return out
            p.sim.evolve_many = spy
            p.sim.batch_steps = 3          # 3 + 3 + 1: the first minimum of EVERY call is global
            p.run_sim()
        res["dts"], res["t"], res["n"] = np.array(dts), p.sim.cc_data.t, p.sim.n
        res["dmax"] = p.sim.cc_data.max("density")
        res["dmin"] = p.sim.cc_data.min("density")
        if case == "comp_write":
            p.sim.write(os.path.join(tmp_out, "out"))
        g = p.get_grid()
        res["rows"] = np.array([g.i0, g.i0 + g.qx])
        res["U"] = np.asarray(p.sim.cc_data.data).copy()
    elif case == "sedov":
        p = Pyro("compressible")
        p.initialize_problem("sedov", inputs_dict={"mesh.nx": 48, "mesh.ny": 32, "driver.max_steps": 5,
                                                   "sedov.r_init": 0.12, "gpu.fast_math": 0})
        res["ic"] = np.asarray(p.sim.cc_data.data).copy()
        p.run_sim()
        g = p.get_grid()
        res["rows"] = np.array([g.i0, g.i0 + g.qx])
        res["U"] = np.asarray(p.sim.cc_data.data).copy()
        res["t"], res["n"] = p.sim.cc_data.t, p.sim.n
    elif case == "refdrv":
        # the REFERENCE's own driver, unmodified (runpy), on the alias package `pyro`: its Pyro
        # class constructs the Simulation, initialize()s and steps it (pyro/pyro_sim.py:182-189,
        # :241-281) -- and under a decomposition every process drives its slab
        import runpy
        import pyro                  # noqa: F401  (this repository's alias package)
        ns = runpy.run_path(REF_DRIVER, run_name="reference_pyro_sim")
        p = ns["Pyro"]("compressible")
        p.initialize_problem("sedov", inputs_dict={"mesh.nx": 64, "mesh.ny": 64, "driver.max_steps": 6})
        dts = []
        while not p.sim.finished():
            p.single_step()
            dts.append(p.sim.dt)
        g = p.get_grid()
        res["dts"] = np.array(dts)
        res["rows"] = np.array([g.i0, g.i0 + g.qx])
        res["U"] = np.asarray(p.sim.cc_data.data).copy()
        res["t"], res["n"] = p.sim.cc_data.t, p.sim.n
    elif case == "adv":
        p = Pyro("advection")
        p.initialize_problem("smooth", inputs_dict={"mesh.nx": 32, "mesh.ny": 16, "driver.max_steps": 9,
                                                    "advection.u": 1.0, "advection.v": -0.5, "particles.do_particles": 0,
                                                    "gpu.fast_math": 0})
        p.run_sim()
        g = p.get_grid()
        res["rows"] = np.array([g.i0, g.i0 + g.qx])
        res["U"] = np.asarray(p.sim.cc_data.data).copy()
        res["t"], res["n"] = p.sim.cc_data.t, p.sim.n
    return res


def _worker(rank, world, port, case, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    os.chdir(out_dir)
    os.makedirs(f"r{rank}", exist_ok=True)
    os.chdir(f"r{rank}")                 # (inputs.auto of rank 0)
    import ctypes as C
    import torch.distributed as td
    td.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import build_emu
    from host_comm import HostStagedComm
    from pyro2_amd import _lib, decomp, device
    _lib.use_library(build_emu.LIB, allow_backends=("host-emu",))
    ctx = device.Context(0)
    device.Context._default = ctx
    host = HostStagedComm(td)
    if case in ("comp_batched", "comp_write", "sedov"):
        # "the communicator lives in the library": pyrohip_halo_exchange and the device-side
        # all-reduce of the CFL minimum are answered over gloo (tests/emu/comm_emu.cpp)
        holder = {}

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)
        def halo_cb(handle, lo, hi):
            host.halo_exchange(holder["state"], lo, hi)
            return 0

        @C.CFUNCTYPE(C.c_double, C.c_double)
        def min_cb(x):
            return host.allreduce_min(x)
        l = _lib.lib()
        l.pyrohip_emu_set_comm.argtypes = [C.c_void_p, C.c_void_p]
        l.pyrohip_emu_set_comm(C.cast(halo_cb, C.c_void_p), C.cast(min_cb, C.c_void_p))

        class LibComm(decomp.RcclComm):
            def halo_exchange(self, state, lo, hi):
                holder["state"] = state
                super().halo_exchange(state, lo, hi)

            def allreduce_min(self, x):
                return host.allreduce_min(x)

            def gather(self, state, dec):
                return host.gather(state, dec)
        comm = LibComm(ctx, global_dt=True)
        orig_evolve = decomp.SlabCompressible.evolve

        def evolve(self, *a, **kw):        # (the callback needs the state before the first exchange)
            holder["state"] = self.state
            return orig_evolve(self, *a, **kw)
        decomp.SlabCompressible.evolve = evolve
    else:
        comm = host
    decomp.set_decomposition(comm, rank, world)
    res = run_case(case, out_dir)
    np.savez(os.path.join(out_dir, f"{case}_{rank}.npz"), **res)
    decomp.set_decomposition(None)
    td.barrier()
    td.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(case, tmp_path, world):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build()
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)


@pytest.fixture
def single(tmp_path, monkeypatch):
    """the same script in THIS process on the emulator, no decomposition"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from pyro2_amd import _lib, decomp, device
    _lib.use_library(build_emu.LIB, allow_backends=("host-emu",))
    ctx = device.Context(0)
    monkeypatch.setattr(device.Context, "_default", ctx)
    decomp.set_decomposition(None)
    d = tmp_path / "single"
    d.mkdir()
    monkeypatch.chdir(d)

    def run(case):
        return run_case(case, str(d))
    return run


def _same_slabs(tmp_path, case, world, want, ng=4):
    seen = 0
    for r in range(world):
        z = np.load(tmp_path / f"{case}_{r}.npz")
        a, b = z["rows"]
        assert np.array_equal(z["U"][ng:-ng, ng:-ng], want["U"][a + ng:b - ng, ng:-ng]), (case, r)
        assert float(z["t"]) == want["t"] and int(z["n"]) == want["n"], (case, r)
        seen += b - a - 2 * ng
    assert seen == want["U"].shape[0] - 2 * ng
    return z


@pytest.mark.parametrize("world", [2, 4])
def test_pyro_compressible_single_steps_decomposed(tmp_path, single, world):
    """Pyro.single_step() x 7 on 2 / 4 slabs (off-centre blast: the ranks' own CFL minima differ
    from the first step on): state, time and the dt of EVERY step as in the single-domain run;
    CellCenterData2d.min / max are the whole grid's"""
    _spawn("comp_steps", tmp_path, world)
    want = single("comp_steps")
    assert len(want["dts"]) == 7 and np.abs(want["U"][4:-4, 4:-4, 2]).max() > 0.1
    for r in range(world):
        z = np.load(tmp_path / f"comp_steps_{r}.npz")
        assert np.array_equal(z["dts"], want["dts"]), (r, z["dts"], want["dts"])
        assert float(z["dmax"]) == float(want["dmax"]) and float(z["dmin"]) == float(want["dmin"])
    _same_slabs(tmp_path, "comp_steps", world, want)


@pytest.mark.parametrize("world", [2, 4])
def test_pyro_compressible_run_sim_batched_decomposed(tmp_path, single, world):
    """Pyro.run_sim() with the steps batched on the device (Simulation.evolve_many ->
    pyrohip_comp_evolve on every slab: halo exchange, ghost fill, all-reduced CFL minimum, dt
    policy and update back to back; 3 + 3 + 1 steps)"""
    _spawn("comp_batched", tmp_path, world)
    want = single("comp_batched")
    assert len(want["dts"]) == 7
    for r in range(world):
        z = np.load(tmp_path / f"comp_batched_{r}.npz")
        assert np.array_equal(z["dts"], want["dts"]), (r, z["dts"], want["dts"])
    _same_slabs(tmp_path, "comp_batched", world, want)


def test_pyro_sedov_problem_decomposed(tmp_path, single):
    """the shipped sedov problem (compressible/problems/sedov.py: sub-sampled blast at the centre,
    i.e. ON the cut between two slabs): initial condition and 5 steps"""
    _spawn("sedov", tmp_path, 2)
    want = single("sedov")
    for r in range(2):
        z = np.load(tmp_path / f"sedov_{r}.npz")
        a, b = z["rows"]
        assert np.array_equal(z["ic"], want["ic"][a:b]), r
    _same_slabs(tmp_path, "sedov", 2, want)


def test_pyro_write_gathers_one_file(tmp_path, single):
    """Simulation.write() of a decomposed run: ONE file in the reference's layout
    (pyro/mesh/patch.py:750-788) written by rank 0, equal to the single-domain run's"""
    from pyro2_amd.util import io_pyro
    _spawn("comp_write", tmp_path, 2)
    single("comp_write")
    mine = io_pyro.read(str(tmp_path / "single" / "out"))
    theirs = io_pyro.read(str(tmp_path / "out"))
    assert not os.path.exists(tmp_path / "r1" / "out.h5")
    g1, g2 = mine.cc_data.grid, theirs.cc_data.grid
    assert (g1.nx, g1.ny, g1.xmin, g1.xmax) == (g2.nx, g2.ny, g2.xmin, g2.xmax)
    assert theirs.cc_data.t == mine.cc_data.t and theirs.n == mine.n
    for name in mine.cc_data.names:
        assert np.array_equal(theirs.cc_data.get_var(name).v(), mine.cc_data.get_var(name).v()), name


@pytest.mark.parametrize("world", [2, 4])
def test_pyro_advection_decomposed(tmp_path, single, world):
    """Pyro("advection").run_sim() on a periodic grid cut into 2 / 4 slabs (the wrap-around is a
    halo too; with two ranks both neighbours are the same process)"""
    _spawn("adv", tmp_path, world)
    want = single("adv")
    _same_slabs(tmp_path, "adv", world, want)


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="no reference checkout here")
def test_reference_driver_runs_decomposed(tmp_path, single, golden):
    """the reference's own pyro_sim.py (unmodified, runpy) on the alias package in TWO processes:
    its Pyro("compressible") steps the sedov problem in two slabs -- the dt of every step and the
    state equal the single-domain run of the same driver bit for bit, and the reference's own
    run (tests/golden/comp_sedov_64_020.npz) to 1e-12"""
    _spawn("refdrv", tmp_path, 2)
    want = single("refdrv")
    g = golden("comp_sedov_64_020")
    assert np.abs(want["dts"] - g["dts"][:6]).max() <= 1e-12 * np.abs(g["dts"][:6]).max()
    for r in range(2):
        z = np.load(tmp_path / f"refdrv_{r}.npz")
        assert np.array_equal(z["dts"], want["dts"]), r
    _same_slabs(tmp_path, "refdrv", 2, want)
