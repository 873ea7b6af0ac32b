"""x-slab domain decomposition of a Grid2d across the GPUs of one node.

The reference is single-block (pyro/mesh/array_indexer.py:157-158 "there is
only a single grid"); the explicit finite-volume update has a dependence
radius of ng = 4 cells, so slabs of whole rows (i is the slow axis: each halo
is ng contiguous rows per variable) with one halo exchange per time step and a
min all-reduce of dt reproduce the single-domain result bit for bit, provided
the artificial-viscosity coefficient is also computed on slab faces that are
interior to the global grid.
"""
import numpy as np

from ._lib import BC_CODE, BC_HALO


class SlabDecomp:
    def __init__(self, nx, nranks, rank, periodic=False):
        if nranks < 1 or not 0 <= rank < nranks:
            raise ValueError("bad rank / nranks")
        base, rem = divmod(int(nx), nranks)
        counts = [base + (1 if r < rem else 0) for r in range(nranks)]
        if min(counts) < 4:
            raise ValueError("slab thinner than the ghost width")
        self.nx, self.nranks, self.rank = int(nx), nranks, rank
        self.counts = counts
        self.nx_local = counts[rank]
        # local array row r  <->  global array row i0 + r (both carry ng ghosts)
        self.i0 = sum(counts[:rank])
        self.lo = rank - 1 if rank > 0 else (nranks - 1 if periodic and nranks > 1 else -1)
        self.hi = rank + 1 if rank < nranks - 1 else (0 if periodic and nranks > 1 else -1)

    def var_bcs(self, per_var):
        """per-variable [xl,xr,yl,yr] names/codes -> int table with HALO on the
        slab faces that have a neighbour"""
        out = np.zeros((len(per_var), 4), dtype=np.int32)
        for n, row in enumerate(per_var):
            for s, b in enumerate(row):
                out[n, s] = BC_CODE[b] if isinstance(b, str) else int(b)
            if self.lo >= 0:
                out[n, 0] = BC_HALO
            if self.hi >= 0:
                out[n, 1] = BC_HALO
        return [list(r) for r in out]

    def comp_var_bcs(self, bcs):
        """mesh boundary names -> table for (dens, ener, xmom, ymom);
        'reflect' is odd for the normal momentum (simulation_null.py:99-112)"""
        rows = []
        for n in range(4):
            row = []
            for s, b in enumerate(bcs):
                if b == "reflect":
                    odd = (n == 2 and s < 2) or (n == 3 and s >= 2)
                    row.append("reflect-odd" if odd else "reflect-even")
                else:
                    row.append(b)
            rows.append(row)
        return self.var_bcs(rows)

    def local_rows(self, ng):
        """global array rows [a, b) held by this rank incl. ghosts"""
        return self.i0, self.i0 + self.nx_local + 2 * ng


class RcclComm:
    """data-path communication of the product: RCCL inside libpyrohip
    (csrc/comm.hip) on the context's stream"""

    overlap = True     # post the next halo exchange beside the interior update

    def __init__(self, ctx, global_dt=True):
        self.ctx = ctx
        # the step kernel's CFL minimum is all-reduced on the device inside
        # comp_step: one host round trip per step less
        ctx.comm_set_global_dt(global_dt)

    def halo_exchange(self, state, lo, hi):
        state.halo_exchange(lo, hi)

    def allreduce_min(self, x):
        return self.ctx.allreduce_min(x)

    def dt_min(self, state, params, cfl):
        """global CFL time step of a decomposed state"""
        if state.comp_dt_is_global():
            return state.comp_dt(params, cfl)
        return self.allreduce_min(state.comp_dt(params, cfl))


class HostStagedComm:
    """same contract as RcclComm over a torch.distributed process group
    (gloo) with the halo rows staged through host memory.  NOT the product
    data path: used by the CPU test-suite (no RCCL without GPUs) and as the
    loudly reported fallback of bench.py when the RCCL communicator cannot
    be created."""

    # with the emulated backend set_neighbours only selects the launch order of the
    # row-marching kernel (boundary strips first): same results, exercised on CPU
    overlap = True

    def __init__(self, td):
        self.td = td

    def halo_exchange(self, state, lo, hi):
        import torch
        ng, nxl = state.ng, state.nx
        reqs, recvs = [], []
        # same pairing as csrc/comm.hip: low rows -> lo, hi ghosts <- hi,
        # high rows -> hi, lo ghosts <- lo
        if lo >= 0:
            t = torch.from_numpy(state.download_rows(ng, ng).copy())
            reqs.append(self.td.isend(t, lo, tag=1))
        if hi >= 0:
            buf = torch.empty((ng, state.qy, state.nvar), dtype=torch.float64)
            reqs.append(self.td.irecv(buf, hi, tag=1))
            recvs.append((nxl + ng, buf))
        if hi >= 0:
            t = torch.from_numpy(state.download_rows(nxl, ng).copy())
            reqs.append(self.td.isend(t, hi, tag=2))
        if lo >= 0:
            buf = torch.empty((ng, state.qy, state.nvar), dtype=torch.float64)
            reqs.append(self.td.irecv(buf, lo, tag=2))
            recvs.append((0, buf))
        for r in reqs:
            r.wait()
        for row, buf in recvs:
            state.upload_rows(row, buf.numpy())

    def allreduce_min(self, x):
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.MIN)
        return float(t[0])


class NoComm:
    """single rank"""

    def halo_exchange(self, state, lo, hi):
        pass

    def allreduce_min(self, x):
        return x


class SlabCompressible:
    """Pyro.single_step for one x-slab of a decomposed compressible run:
    halo exchange -> y / physical ghost fill -> global CFL dt -> evolve.

    The driver's dt policy (simulation_null.py:222-244) is applied to the
    GLOBAL minimum, so every rank takes the same step."""

    def __init__(self, ctx, decomp, ny, bcs, params_kw, comm, ng=4):
        from . import device
        if any(b in ("hse", "ambient") for b in bcs):
            # the hse energy fill reads the momenta's x ghosts of the PREVIOUS
            # fill (fill_BC_all order); a halo exchange refreshes all variables
            # at once, so the corner ghosts would differ from the single-domain run
            raise NotImplementedError("hse / ambient boundaries are single-GPU only")
        self.dec, self.comm = decomp, comm
        self.state = device.DeviceState(ctx, decomp.nx_local, ny, ng, decomp.comp_var_bcs(bcs))
        kw = dict(params_kw)
        kw["avisc_xhi_interior"] = int(decomp.hi >= 0)
        self.params = device.make_comp_params(**kw)
        # boundary strips first + halo exchange beside the interior strips (kernel_set 2)
        if getattr(comm, "overlap", False) and (decomp.lo >= 0 or decomp.hi >= 0):
            self.state.set_neighbours(decomp.lo, decomp.hi)

    def evolve(self, policy, cfl, nsteps):
        """nsteps of step() enqueued on the device without a host round trip per step
        (pyrohip_comp_evolve: halo exchange, ghost fill, dt policy and update kernels
        back to back; one synchronisation at the end).  Needs the communication inside
        the library (single rank or RcclComm)."""
        if not isinstance(self.comm, (NoComm, RcclComm)):
            raise NotImplementedError("device-side stepping needs RCCL (or a single rank)")
        if isinstance(self.comm, RcclComm) and (self.dec.lo >= 0 or self.dec.hi >= 0):
            self.state.set_neighbours(self.dec.lo, self.dec.hi)
        return self.state.comp_evolve(self.params, cfl, policy, nsteps)

    def step(self, policy, cfl):
        self.comm.halo_exchange(self.state, self.dec.lo, self.dec.hi)
        self.state.fill_bc()
        if hasattr(self.comm, "dt_min"):
            dt = policy(self.comm.dt_min(self.state, self.params, cfl))
        else:
            dt = policy(self.comm.allreduce_min(self.state.comp_dt(self.params, cfl)))
        self.state.comp_step(self.params, dt)
        policy.advance(dt)
        return dt


class DtPolicy:
    """driver time-step policy, pyro/simulation_null.py:222-244: first step
    scaled by init_tstep_factor, growth capped by max_dt_change, last step
    clipped to land on tmax"""

    def __init__(self, tmax, init_tstep_factor=0.01, max_dt_change=2.0, fix_dt=-1.0):
        self.tmax, self.f0, self.mx, self.fix = tmax, init_tstep_factor, max_dt_change, fix_dt
        self.n, self.t, self.dt_old = 0, 0.0, -1.e33

    def __call__(self, dt_method):
        if self.fix > 0.0:
            dt = self.fix
        else:
            dt = dt_method if self.n else self.f0 * dt_method
            if self.n:
                dt = min(self.mx * self.dt_old, dt)
            self.dt_old = dt
        if self.t + dt > self.tmax:
            dt = self.tmax - self.t
        return dt

    def advance(self, dt):
        self.t += dt
        self.n += 1
