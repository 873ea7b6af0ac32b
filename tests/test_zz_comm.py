"""RCCL plumbing on ONE GPU: a 1-rank communicator whose both neighbours are
the rank itself must reproduce the periodic x ghost fill (send/recv to self
inside one group), and the scalar all-reduce must return its input.  The
N > 1 logic is covered on CPU by tests/test_decomp_gloo.py (gloo)."""
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
