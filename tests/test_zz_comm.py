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
