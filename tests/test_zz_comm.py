"""RCCL plumbing on ONE GPU: a 1-rank communicator whose both neighbours are
the rank itself must reproduce the periodic x ghost fill (send/recv to self
inside one group), and the scalar all-reduce must return its input.  The
N > 1 logic is covered on CPU by tests/test_decomp_gloo.py (gloo)."""
import os

import numpy as np
import pytest

from pyro2_amd import device


@pytest.mark.gpu
def test_rccl_self_halo_equals_periodic_fill(hip):
    uid = device.Context.comm_unique_id()
    assert len(uid) == 128
    hip.comm_init(1, 0, uid)
    rng = np.random.default_rng(5)
    nx, ny, ng = 40, 24, 4
    a = rng.standard_normal((nx + 2 * ng, ny + 2 * ng, 4))
    per = device.DeviceState(hip, nx, ny, ng, [["periodic"] * 4] * 4)
    per.upload(a)
    per.fill_bc()
    ref = per.download()
    s = device.DeviceState(hip, nx, ny, ng, [["halo", "halo", "periodic", "periodic"]] * 4)
    s.upload(a)
    s.halo_exchange(0, 0)
    s.fill_bc()
    assert np.array_equal(s.download(), ref)
    assert hip.allreduce_min(3.25) == 3.25
    assert hip.allreduce_max(-1.5) == -1.5


@pytest.mark.gpu
def test_rccl_step_with_device_side_dt_allreduce(hip):
    """SlabCompressible over a 1-rank RCCL communicator (periodic in x through
    self-neighbours): the CFL minimum is all-reduced on the device inside
    comp_step; results identical to the plain single-domain run"""
    from pyro2_amd.decomp import DtPolicy, NoComm, RcclComm, SlabCompressible, SlabDecomp
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from sedov_ic import sedov_ic
    try:      # the context may already carry the 1-rank communicator of the test above
        hip.comm_init(1, 0, device.Context.comm_unique_id())
    except Exception:
        pass
    nx = 128
    ic, meta, bcs = sedov_ic(nx, r_init=0.05)
    kw = dict(dx=1.0 / nx, dy=1.0 / nx, fast_math=0, kernel_set=1)
    runs = []
    for comm in (NoComm(), RcclComm(hip)):
        dec = SlabDecomp(nx, 1, 0, periodic=False)
        slab = SlabCompressible(hip, dec, nx, ["outflow"] * 4, kw, comm)
        slab.state.upload(ic)
        pol = DtPolicy(tmax=0.1)
        dts = [slab.step(pol, 0.8) for _ in range(6)]
        if isinstance(comm, RcclComm):
            assert slab.state.comp_dt_is_global()
        runs.append((slab.state.download(), dts))
    hip.comm_set_global_dt(False)
    assert runs[0][1] == runs[1][1]
    assert np.array_equal(runs[0][0], runs[1][0])


@pytest.mark.gpu
@pytest.mark.parametrize("fast", [0, 1])
def test_rccl_overlapped_slab_short_strips_at_the_end_of_the_interior(hip, fast):
    """a slab whose launch is four and more rounds of wavefronts ends its INTERIOR launch with short strips (the
    row strips in front of the last boundary strip cut in two, dealt to the ends of the XCD queues:
    comp_wave.hip).  1792 x 4096 cells in 16-row strips = 74 x 112 = 8288 wavefronts on 2048 slots -> 83 long
    row strips, 56 short ones, the last boundary strip.  Both x neighbours the rank itself = the periodic
    single-domain run (which cuts ITS last strips short another way): 3 steps, host-stepped and enqueued on
    the device, bit for bit in both builds (a cell is computed the same way whatever strip it sits in)"""
    from pyro2_amd.decomp import DtPolicy
    try:
        hip.comm_init(1, 0, device.Context.comm_unique_id())
    except Exception:
        pass
    hip.comm_set_global_dt(False)
    nx, ny, ng = 1792, 4096, 4
    rng = np.random.default_rng(5)
    x = (np.arange(nx + 2 * ng)[:, None] - ng + 0.5) / nx
    y = (np.arange(ny + 2 * ng)[None, :] - ng + 0.5) / ny
    full = np.zeros((nx + 2 * ng, ny + 2 * ng, 4))
    a = rng.uniform(0.05, 0.2, 6)
    rho = 1.0 + a[0] * np.sin(2 * np.pi * x) * np.cos(4 * np.pi * y) + a[1] * np.cos(6 * np.pi * x)
    u = a[2] * np.sin(4 * np.pi * x) + a[3] * np.cos(2 * np.pi * y)
    v = a[4] * np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y)
    pr = 1.0 + a[5] * np.cos(2 * np.pi * x) * np.sin(6 * np.pi * y)
    full[..., 0] = rho
    full[..., 2] = rho * u
    full[..., 3] = rho * v
    full[..., 1] = pr / 0.4 + 0.5 * rho * (u * u + v * v)
    P = device.make_comp_params(1.0 / nx, 1.0 / ny, fast_math=fast, kernel_set=2, march_rows=16)

    def run(mode, on_device):
        bx = "periodic" if mode == "periodic" else "halo"
        s = device.DeviceState(hip, nx, ny, ng, [[bx, bx, "outflow", "outflow"]] * 4)
        s.upload(full)
        if mode == "overlap":
            s.set_neighbours(0, 0)
        pol, dts = DtPolicy(0.1), []
        if on_device:
            dts = list(s.comp_evolve(P, 0.8, pol, 3))
        else:
            for _ in range(3):
                if mode != "periodic":
                    s.halo_exchange(0, 0)
                s.fill_bc()
                dt = pol(s.comp_dt(P, 0.8))
                s.comp_step(P, dt)
                pol.advance(dt)
                dts.append(dt)
        return s.download()[ng:-ng, ng:-ng], dts

    ref, dref = run("periodic", False)
    for on_device in (False, True):
        U, d = run("overlap", on_device)
        assert d == dref, on_device
        assert np.array_equal(U, ref), on_device


@pytest.mark.gpu
def test_rccl_overlapped_halo_self_neighbour(hip):
    """the overlapped exchange (boundary strips first, halos of the NEW state on the
    halo stream / second communicator beside the interior strips, next step only
    waits) on ONE GPU: a slab whose both x neighbours are the rank itself is the
    periodic single-domain problem.  Sedov blast crossing the periodic x boundary,
    kernel_set 2 with 16-row strips (8 strips), 12 steps: bit-identical to the run
    with periodic boundaries and no communicator -- and to the synchronous
    exchange."""
    from pyro2_amd.decomp import DtPolicy
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from sedov_ic import sedov_ic
    try:
        hip.comm_init(1, 0, device.Context.comm_unique_id())
    except Exception:
        pass
    hip.comm_set_global_dt(False)
    nx, ng = 128, 4
    ic, meta, _ = sedov_ic(nx, r_init=0.05)
    ic = np.roll(np.nan_to_num(ic)[ng:-ng, ng:-ng], nx // 2 - 6, axis=0)     # blast next to the x boundary
    full = np.zeros((nx + 2 * ng, nx + 2 * ng, 4))
    full[ng:-ng, ng:-ng] = ic
    P = device.make_comp_params(1.0 / nx, 1.0 / nx, fast_math=0, kernel_set=2, march_rows=16)

    def run(mode):
        bx = "periodic" if mode == "periodic" else "halo"
        s = device.DeviceState(hip, nx, nx, ng, [[bx, bx, "outflow", "outflow"]] * 4)
        s.upload(full)
        if mode == "overlap":
            s.set_neighbours(0, 0)
        pol, dts = DtPolicy(0.1), []
        for _ in range(12):
            if mode != "periodic":
                s.halo_exchange(0, 0)
            s.fill_bc()
            dt = pol(s.comp_dt(P, 0.8))
            s.comp_step(P, dt)
            assert s.halo_pending() == (mode == "overlap")     # the step posted the next exchange
            pol.advance(dt)
            dts.append(dt)
        return s.download()[ng:-ng, ng:-ng], dts

    ref, dref = run("periodic")
    for mode in ("sync", "overlap"):
        U, d = run(mode)
        assert d == dref, mode
        assert np.array_equal(U, ref), mode
    # the same 12 steps enqueued on the device in one call (pyrohip_comp_evolve): halo
    # exchange beside the interior strips, ghost fill, dt policy kernel, update -- no host
    # round trip in between
    s = device.DeviceState(hip, nx, nx, ng, [["halo", "halo", "outflow", "outflow"]] * 4)
    s.upload(full)
    s.set_neighbours(0, 0)
    pol = DtPolicy(0.1)
    d = list(s.comp_evolve(P, 0.8, pol, 12))
    assert d == dref
    assert np.array_equal(s.download()[ng:-ng, ng:-ng], ref)
    assert np.abs(ref[:4, :, 2]).max() > 0.0          # the blast did reach the x boundary
    # ... and with the tile kernel, which from the second step on reads its ghost cells
    # through the boundary rules (fuse_fill): halo rows are data, the y sides outflow
    P1 = device.make_comp_params(1.0 / nx, 1.0 / nx, fast_math=0, kernel_set=1)
    s = device.DeviceState(hip, nx, nx, ng, [["halo", "halo", "outflow", "outflow"]] * 4)
    s.upload(full)
    s.set_neighbours(0, 0)
    pol = DtPolicy(0.1)
    d = list(s.comp_evolve(P1, 0.8, pol, 12))
    assert d == dref
    assert np.array_equal(s.download()[ng:-ng, ng:-ng], ref)


def _rank_device(dv, rank, world):
    """GPU of a rank.  On a box with fewer GPUs than ranks the ranks share them: RCCL refuses
    two ranks on one device of one host ("Duplicate GPU detected"), so each rank names itself a
    host of its own (NCCL_HOSTID) and the communicator runs over RCCL's socket transport on lo
    -- the same RCCL calls, streams and events as between two GPUs, none of the bandwidth."""
    import os
    ndev = dv.device_count()
    if ndev < world:
        os.environ["NCCL_HOSTID"] = f"pyro2amd-test-rank{rank}"
        os.environ["NCCL_IB_DISABLE"] = "1"
        os.environ["NCCL_P2P_DISABLE"] = "1"
    return rank % ndev


def _run_two_ranks(fn, args, timeout=240.0, nprocs=2):
    """two ranks of `fn`, bounded in time (a communicator that does not come up must not hang
    the suite): None, or what went wrong.  With one GPU that is a skip, with two a failure."""
    import time
    import torch.multiprocessing as mp
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    t0 = time.time()
    err = None
    try:
        while not ctx.join(timeout=5.0):
            if time.time() - t0 > timeout:
                err = f"no result after {timeout:.0f} s"
                break
    except Exception as e:      # noqa: BLE001  (a rank raised: torch re-raises it here)
        err = f"{type(e).__name__}: {str(e)[-400:]}"
    if err is not None:
        for pr in ctx.processes:
            if pr.is_alive():
                pr.kill()
        if device.device_count() < nprocs:
            pytest.skip("RCCL ranks sharing a GPU (socket transport) did not run here: " + err)
        pytest.fail(err)


def _rccl_rank(rank, world, port, out_dir):
    """one rank of the 2-GPU test below"""
    import os
    import sys
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME="lo")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    from pyro2_amd import device as dv
    from pyro2_amd.decomp import DtPolicy, RcclComm, SlabCompressible, SlabDecomp
    from sedov_ic import sedov_ic
    ctx = dv.Context(_rank_device(dv, rank, world))
    t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        t = torch.frombuffer(bytearray(dv.Context.comm_unique_id()), dtype=torch.uint8).clone()
    td.broadcast(t, 0)
    ctx.comm_init(world, rank, bytes(t.numpy().tobytes()))
    nx = 256
    ic, meta, bcs = sedov_ic(nx, r_init=0.05)
    dec = SlabDecomp(nx, world, rank)
    kw = dict(dx=1.0 / nx, dy=1.0 / nx, fast_math=0, kernel_set=2, march_rows=16)
    sl = SlabCompressible(ctx, dec, nx, bcs, kw, RcclComm(ctx))
    a, b = dec.local_rows(4)
    sl.state.upload(np.nan_to_num(ic)[a:b])
    pol = DtPolicy(0.1)
    dts = [sl.step(pol, 0.8) for _ in range(20)]
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), U=sl.state.download(), dts=np.array(dts),
             rows=np.array([a, b]))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.gpu
def test_rccl_two_ranks_bit_identical(hip, tmp_path):
    """2 GPUs, one process each, x slabs exchanged over RCCL (overlapped with the
    interior update) and the dt all-reduced on the device: bit-identical to the
    single-GPU run.  On a box with one GPU the two ranks share it and RCCL runs
    over its socket transport (_rank_device): the protocol is the same."""
    import socket
    import sys, os
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(__file__))
    from sedov_ic import sedov_ic
    from pyro2_amd.decomp import DtPolicy
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    _run_two_ranks(_rccl_rank, (2, port, str(tmp_path)))
    nx, ng = 256, 4
    ic, meta, bcs = sedov_ic(nx, r_init=0.05)
    s = device.DeviceState(hip, nx, nx, ng, [["outflow"] * 4] * 4)
    s.upload(np.nan_to_num(ic))
    P = device.make_comp_params(1.0 / nx, 1.0 / nx, fast_math=0, kernel_set=2, march_rows=16)
    pol, dts = DtPolicy(0.1), []
    for _ in range(20):
        s.fill_bc()
        dt = pol(s.comp_dt(P, 0.8))
        s.comp_step(P, dt)
        pol.advance(dt)
        dts.append(dt)
    ref = s.download()
    for r in range(2):
        d = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert list(d["dts"]) == dts, r
        a, b = d["rows"]
        assert np.array_equal(d["U"][ng:-ng, ng:-ng], ref[a + ng:b - ng, ng:-ng]), r


def _rccl_evolve_rank(rank, world, port, out_dir):
    """one rank of the four-rank test below"""
    import os
    import sys
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME="lo")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    from pyro2_amd import device as dv
    from pyro2_amd.decomp import DtPolicy, RcclComm, SlabCompressible, SlabDecomp
    from sedov_ic import sedov_ic
    ctx = dv.Context(_rank_device(dv, rank, world))
    t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        t = torch.frombuffer(bytearray(dv.Context.comm_unique_id()), dtype=torch.uint8).clone()
    td.broadcast(t, 0)
    ctx.comm_init(world, rank, bytes(t.numpy().tobytes()))
    nx = 256
    ic, meta, bcs = sedov_ic(nx, r_init=0.05)
    dec = SlabDecomp(nx, world, rank)
    out = {}
    for fm in (0, 1):
        kw = dict(dx=1.0 / nx, dy=1.0 / nx, fast_math=fm, kernel_set=2, march_rows=16)
        sl = SlabCompressible(ctx, dec, nx, bcs, kw, RcclComm(ctx))
        a, b = dec.local_rows(4)
        sl.state.upload(np.nan_to_num(ic)[a:b])
        pol = DtPolicy(0.1)
        dts = list(sl.evolve(pol, 0.8, 7)) + [sl.step(pol, 0.8) for _ in range(2)] + \
            list(sl.evolve(pol, 0.8, 5))
        out[f"U{fm}"] = sl.state.download()
        out[f"dts{fm}"] = np.array(dts)
        out[f"t{fm}"] = np.array(pol.t)
        del sl
    np.savez(os.path.join(out_dir, f"e{rank}.npz"), rows=np.array([a, b]), **out)
    td.barrier()
    td.destroy_process_group()


@pytest.mark.gpu
def test_rccl_four_ranks_device_side_stepping(hip, tmp_path):
    """FOUR ranks (sharing the box's GPUs: _rank_device), device-side stepping
    (pyrohip_comp_evolve) in two calls with host-side steps between them, both builds: every
    rank takes the dt of the GLOBAL CFL minimum -- also in the first step of a call, where the
    minimum comes from a kernel of its own (comp_cfl_min_device) -- and the slabs equal the
    single-domain run bit for bit.  With two ranks the blast sits on the cut and both local
    minima are equal: the outer slabs of four see ambient gas only, their own minimum is 40 x
    larger (found on hardware in round 4, profiles/r04_rccl_ranks_one_gpu.txt)."""
    import socket
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from sedov_ic import sedov_ic
    from pyro2_amd.decomp import DtPolicy
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    _run_two_ranks(_rccl_evolve_rank, (4, port, str(tmp_path)), timeout=300.0, nprocs=4)
    nx, ng = 256, 4
    ic, meta, bcs = sedov_ic(nx, r_init=0.05)
    for fm in (0, 1):
        s = device.DeviceState(hip, nx, nx, ng, [["outflow"] * 4] * 4)
        s.upload(np.nan_to_num(ic))
        P = device.make_comp_params(1.0 / nx, 1.0 / nx, fast_math=fm, kernel_set=2, march_rows=16)
        pol = DtPolicy(0.1)
        dts = list(s.comp_evolve(P, 0.8, pol, 14))
        ref = s.download()
        for r in range(4):
            d = np.load(os.path.join(str(tmp_path), f"e{r}.npz"))
            assert list(d[f"dts{fm}"]) == dts, (fm, r)
            assert float(d[f"t{fm}"]) == pol.t, (fm, r)
            a, b = d["rows"]
            assert np.array_equal(d[f"U{fm}"][ng:-ng, ng:-ng], ref[a + ng:b - ng, ng:-ng]), (fm, r)


def _periodic_ic(nx):
    """Sedov blast moved off the centre onto the wrap-around cut of a grid periodic in x, plus
    a smooth momentum field (no symmetry between the two slabs)"""
    from sedov_ic import sedov_ic
    ic, meta, _ = sedov_ic(nx, r_init=0.05)
    U = np.nan_to_num(ic)[4:-4, 4:-4].copy()
    U = np.roll(U, nx // 2 - 9, axis=0)
    i = (np.arange(nx) + 0.5) / nx
    U[:, :, 2] += 1.0e-2 * U[:, :, 0] * np.sin(2 * np.pi * i)[:, None]
    U[:, :, 1] += 0.5 * 1.0e-4 * U[:, :, 0] * np.sin(2 * np.pi * i)[:, None] ** 2
    full = np.zeros((nx + 8, nx + 8, 4))
    full[4:-4, 4:-4] = U
    return full


def _rccl_periodic_rank(rank, world, port, out_dir):
    """one rank of the periodic two-rank test below"""
    import os
    import sys
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME="lo")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    from pyro2_amd import device as dv
    from pyro2_amd.decomp import DtPolicy, RcclComm, SlabCompressible, SlabDecomp
    ctx = dv.Context(_rank_device(dv, rank, world))
    t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        t = torch.frombuffer(bytearray(dv.Context.comm_unique_id()), dtype=torch.uint8).clone()
    td.broadcast(t, 0)
    ctx.comm_init(world, rank, bytes(t.numpy().tobytes()))
    nx = 128
    full = _periodic_ic(nx)
    dec = SlabDecomp(nx, world, rank, periodic=True)
    kw = dict(dx=1.0 / nx, dy=1.0 / nx, fast_math=0, kernel_set=2, march_rows=16)
    sl = SlabCompressible(ctx, dec, nx, ["periodic", "periodic", "outflow", "outflow"], kw, RcclComm(ctx))
    a, b = dec.local_rows(4)
    sl.state.upload(full[a:b])
    pol = DtPolicy(0.1)
    dts = [sl.step(pol, 0.8) for _ in range(3)] + list(sl.evolve(pol, 0.8, 9))
    np.savez(os.path.join(out_dir, f"p{rank}.npz"), U=sl.state.download(), dts=np.array(dts),
             rows=np.array([a, b]))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.gpu
def test_rccl_two_ranks_periodic_in_x(hip, tmp_path):
    """two ranks on a grid periodic in x: BOTH neighbours of a rank are the other rank, and the
    four transfers of an exchange between the pair are matched by their order alone (RCCL has no
    tags; comm.hip: send low rows, receive high ghosts, send high rows, receive low ghosts).  A
    blast on the wrap-around cut, host-side steps and device-side stepping: bit-identical to
    the single-domain periodic run."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    _run_two_ranks(_rccl_periodic_rank, (2, port, str(tmp_path)))
    nx, ng = 128, 4
    full = _periodic_ic(nx)
    from pyro2_amd.decomp import DtPolicy
    s = device.DeviceState(hip, nx, nx, ng, [["periodic", "periodic", "outflow", "outflow"]] * 4)
    s.upload(full)
    P = device.make_comp_params(1.0 / nx, 1.0 / nx, fast_math=0, kernel_set=2, march_rows=16)
    pol = DtPolicy(0.1)
    dts = list(s.comp_evolve(P, 0.8, pol, 12))
    ref = s.download()
    assert np.abs(ref[ng:ng + 4, ng:-ng, 2]).max() > 1.0e-3      # the blast did cross the cut
    for r in range(2):
        d = np.load(os.path.join(str(tmp_path), f"p{r}.npz"))
        assert list(d["dts"]) == dts, r
        a, b = d["rows"]
        assert np.array_equal(d["U"][ng:-ng, ng:-ng], ref[a + ng:b - ng, ng:-ng]), r


def _rccl_mg_rank(rank, world, port, out_dir):
    """one rank of the 2-GPU multigrid test below"""
    import os
    import sys
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME="lo")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    from pyro2_amd import device as dv
    from pyro2_amd.multigrid.slab import RcclRowComm, SlabMG
    ctx = dv.Context(_rank_device(dv, rank, world))
    t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        t = torch.frombuffer(bytearray(dv.Context.comm_unique_id()), dtype=torch.uint8).clone()
    td.broadcast(t, 0)
    ctx.comm_init(world, rank, bytes(t.numpy().tobytes()))
    d = np.load(os.path.join(out_dir, "in.npz"))
    nx = d["rhs"].shape[0] - 2
    m = dv.DeviceMG(ctx, nx)
    L = m.nlevels - 1
    m.set(L, 0, d["v0"])
    m.set(L, 1, d["rhs"])
    sm = SlabMG(m, RcclRowComm(rank, world), rank, world, collapse_n=256)
    for _ in range(2):
        sm.vcycle()
    np.savez(os.path.join(out_dir, f"mg{rank}.npz"), v=sm.solution_rows(), rows=np.array(sm.rows(L)))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_rccl_multigrid_slabs_two_ranks(hip, tmp_path, world):
    """2 (4: middle slabs with two neighbours) ranks: Poisson 2048^2 V-cycles, levels 2048^2..512^2 in x slabs with their halo
    rows over RCCL, 256^2 and below collapsed onto rank 0 -- bit-identical to the
    single-GPU V-cycles.  On a box with one GPU the two ranks share it (_rank_device; the same
    SlabMG also runs with host-staged rows in tests/test_device_multigrid.py and over gloo in
    tests/test_decomp_gloo.py)."""
    import socket
    import torch.multiprocessing as mp
    nx = 2048
    x = (np.arange(nx + 2) - 0.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    rhs = -2.0 * ((1 - 6 * X**2) * Y**2 * (1 - Y**2) + (1 - 6 * Y**2) * X**2 * (1 - X**2))
    v0 = np.zeros((nx + 2, nx + 2))
    v0[1:-1, 1:-1] = 0.01 * np.random.default_rng(3).standard_normal((nx, nx))
    np.savez(os.path.join(str(tmp_path), "in.npz"), v0=v0, rhs=rhs)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    _run_two_ranks(_rccl_mg_rank, (world, port, str(tmp_path)), nprocs=world)
    m = device.DeviceMG(hip, nx)
    L = m.nlevels - 1
    m.set(L, 0, v0)
    m.set(L, 1, rhs)
    for _ in range(2):
        for l in range(L):
            m.mark_zero(l)
        m.vcycle(L)
    ref = m.get(L, 0)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"mg{r}.npz"))
        a, b = d["rows"]
        assert np.array_equal(d["v"][:, 1:-1], ref[a:b + 1, 1:-1]), r


@pytest.mark.gpu
def test_rccl_multigrid_row_moves_to_self(hip):
    """the RCCL row moves of the slab V-cycle on ONE GPU: with the rank as its own low
    (high) neighbour the halo rows below (above) the slab become copies of the slab's
    first (last) rows, and a grouped send + recv to self copies a block of rows"""
    from pyro2_amd._lib import check, lib
    try:      # the context may already carry the 1-rank communicator of the tests above
        hip.comm_init(1, 0, device.Context.comm_unique_id())
    except Exception:
        pass
    nx = 256
    m = device.DeviceMG(hip, nx)
    L = m.nlevels - 1
    a = np.random.default_rng(5).standard_normal((nx + 2, nx + 2))
    for var in (0, 1):
        m.set(L, var, a)
        m._call("pyrohip_mg_exchange_rows", L, var, 65, 128, 10, 0, -1)
        got = m.get(L, var)
        want = a.copy()
        want[55:65] = a[65:75]
        assert np.array_equal(got, want)
        m.set(L, var, a)
        m._call("pyrohip_mg_exchange_rows", L, var, 65, 128, 6, -1, 0)
        want = a.copy()
        want[129:135] = a[123:129]
        assert np.array_equal(m.get(L, var), want)
    m.set(L, 0, a)
    check(lib().pyrohip_comm_group(1))
    m._call("pyrohip_mg_send_rows", L, 0, 3, 20, 0)
    m._call("pyrohip_mg_recv_rows", L, 0, 100, 20, 0)
    check(lib().pyrohip_comm_group(0))
    want = a.copy()
    want[100:120] = a[3:23]
    assert np.array_equal(m.get(L, 0), want)


@pytest.mark.gpu
def test_rccl_slab_modified_between_steps(hip):
    """a slab that is written between two host-driven steps while the overlapped
    exchange of its last step is posted.  SlabCompressible.modified() / upload_rows() (collective)
    drop the posted exchange, the next step exchanges synchronously and switches the overlap on
    again: one rank that is its own neighbour on both sides (a periodic single-domain problem),
    Sedov next to the x boundary, 6 steps, the boundary rows scaled from the host, 6 more steps
    -- bit-identical to the same sequence on a periodic state without a communicator; writing
    the state behind the library's back is refused with a message, not a hang."""
    from pyro2_amd.decomp import DtPolicy, RcclComm, SlabCompressible, SlabDecomp
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from sedov_ic import sedov_ic
    try:
        hip.comm_init(1, 0, device.Context.comm_unique_id())
    except Exception:
        pass
    nx, ng = 128, 4
    ic, meta, _ = sedov_ic(nx, r_init=0.05)
    ic = np.roll(np.nan_to_num(ic)[ng:-ng, ng:-ng], nx // 2 - 6, axis=0)
    full = np.zeros((nx + 2 * ng, nx + 2 * ng, 4))
    full[ng:-ng, ng:-ng] = ic
    kw = dict(dx=1.0 / nx, dy=1.0 / nx, fast_math=0, kernel_set=2, march_rows=16)

    class SelfDecomp(SlabDecomp):          # one rank, periodic in x: both neighbours are rank 0
        def __init__(self):
            super().__init__(nx, 1, 0, periodic=False)
            self.lo = self.hi = 0
            self.wrap_lo = self.wrap_hi = True

    def poke(rows):
        rows = rows.copy()
        rows[..., 0] *= 1.25
        rows[..., 1] *= 1.25
        return rows

    # reference: periodic boundaries, no communicator
    P = device.make_comp_params(**kw)
    s = device.DeviceState(hip, nx, nx, ng, [["periodic", "periodic", "outflow", "outflow"]] * 4)
    s.upload(full)
    pol, dref = DtPolicy(0.1), []
    for n in range(12):
        if n == 6:
            s.upload_rows(ng, poke(s.download_rows(ng, 8)))
        s.fill_bc()
        dt = pol(s.comp_dt(P, 0.8))
        s.comp_step(P, dt)
        pol.advance(dt)
        dref.append(dt)
    ref = s.download()[ng:-ng, ng:-ng]

    hip.comm_set_global_dt(False)
    sl = SlabCompressible(hip, SelfDecomp(), nx, ["periodic", "periodic", "outflow", "outflow"], kw,
                          RcclComm(hip, global_dt=False))
    sl.state.upload(full)
    pol, dts = DtPolicy(0.1), []
    for n in range(12):
        if n == 6:
            assert sl.state.halo_pending()
            sl.upload_rows(ng, poke(sl.state.download_rows(ng, 8)))
            assert not sl.state.halo_pending()
        dts.append(sl.step(pol, 0.8))
        assert sl.state.halo_pending()               # the overlap is on (again)
    assert dts == dref
    assert np.array_equal(sl.state.download()[ng:-ng, ng:-ng], ref)
    # the write behind the library's back: refused, with the remedy in the message
    sl.state.upload_rows(ng, sl.state.download_rows(ng, 4))
    with pytest.raises(Exception, match="set_neighbours"):
        sl.step(pol, 0.8)


def _launch_pyro_sim(nproc, port, args, cwd, timeout=420.0):
    """`python -m torch.distributed.run --nproc-per-node N -m pyro2_amd.pyro_sim ...`: the
    launcher line of the driver's multi-GPU bench with pyro's own command line behind it.  The
    workers never import torch: the RCCL unique id travels through pyro2_amd.decomp's own
    rendezvous.  On a box with fewer GPUs than ranks they share (PYRO_SHARE_GPUS=1: RCCL over
    sockets, flagged)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    if nproc > 1:
        if device.device_count() < nproc:
            env["PYRO_SHARE_GPUS"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "pyro2_amd.pyro_sim"]
    else:
        cmd = [sys.executable, "-m", "pyro2_amd.pyro_sim"]
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
    try:
        r = subprocess.run(cmd + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return f"no result after {timeout:.0f} s"
    return None if r.returncode == 0 else (r.stdout[-1500:] + r.stderr[-2500:])


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,batched", [(2, 0), (4, 1)])
def test_pyro_sim_command_line_decomposed_over_rccl(hip, tmp_path, nproc, batched):
    """pyro's command line under the launcher runs ONE decomposed problem --
    `torch.distributed.run --nproc-per-node N -m pyro2_amd.pyro_sim compressible sedov
    inputs.sedov ...` (N processes, x slabs, RCCL halo exchange, global CFL minimum, the output
    file gathered to rank 0) writes the same file, bit for bit, as the plain single-process
    command.  batched = 0: output every step count => Pyro.single_step; 1: nothing between the
    steps => Simulation.evolve_many (device-side stepping of the slabs)."""
    import socket
    from pyro2_amd.util import io_pyro
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    common = ["compressible", "sedov", "inputs.sedov", "mesh.nx=256", "mesh.ny=192", "driver.max_steps=14",
              "gpu.fast_math=0", "vis.dovis=0", "driver.verbose=0"]
    if batched:
        common += ["io.do_io=0", "io.force_final_output=1"]
    else:
        common += ["io.do_io=1", "io.n_out=7", "io.dt_out=1000.0"]
    err = _launch_pyro_sim(nproc, port, common + [f"io.basename={tmp_path}/dec_"], str(tmp_path))
    if err is not None:
        if device.device_count() < nproc:
            pytest.skip("RCCL ranks sharing a GPU (socket transport) did not run here: " + err[-600:])
        pytest.fail(err)
    one = tmp_path / "one"
    one.mkdir()
    err = _launch_pyro_sim(1, port, common + [f"io.basename={tmp_path}/one_"], str(one))
    assert err is None, err
    names = sorted(f for f in os.listdir(tmp_path) if f.startswith("one_"))
    assert names and names[-1].startswith("one_0014"), names
    for f in names:
        a = io_pyro.read(str(tmp_path / f))
        b = io_pyro.read(str(tmp_path / f.replace("one_", "dec_")))
        assert a.cc_data.t == b.cc_data.t and a.n == b.n, f
        assert (b.cc_data.grid.nx, b.cc_data.grid.ny) == (256, 192)
        for v in a.cc_data.names:
            assert np.array_equal(a.cc_data.get_var(v).v(), b.cc_data.get_var(v).v()), (f, v)
    assert np.abs(a.cc_data.get_var("x-momentum").v()).max() > 0.0
