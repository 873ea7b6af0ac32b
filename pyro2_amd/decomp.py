"""x-slab domain decomposition of a Grid2d across the GPUs of one node.

The reference is single-block (pyro/mesh/array_indexer.py:157-158 "there is
only a single grid"); the explicit finite-volume update has a dependence
radius of ng = 4 cells, so slabs of whole rows (i is the slow axis: each halo
is ng contiguous rows per variable) with one halo exchange per time step and a
min all-reduce of dt reproduce the single-domain result bit for bit, provided
the artificial-viscosity coefficient is also computed on slab faces that are
interior to the global grid.
"""
import os
import time

import numpy as np

from ._lib import BC_CODE, BC_HALO


# ---- the decomposition the class surface uses -----------------------------------------------
# pyro's driver (pyro/pyro_sim.py:182-189) builds ONE Simulation per process.  Under a launcher
# that starts one process per GPU (RANK / WORLD_SIZE / LOCAL_RANK in the environment, e.g.
# `python -m torch.distributed.run --nproc-per-node 8 -m pyro2_amd.pyro_sim compressible sedov
# inputs.sedov`) every process then builds the Simulation of ITS x-slab of the one problem:
# grid_setup() hands out the slab's Grid2d, CellCenterData2d exchanges halo rows in fill_BC_all,
# compute_timestep takes the global CFL minimum, write() gathers -- see simulation_null.py,
# mesh/patch.py.  What is in force is kept here.

class Decomposition:
    """rank / nranks of this process and how to get the object that moves halo rows
    (`comm_for(ctx)`: RcclComm in the product, a gloo transport in the CPU tests)"""

    def __init__(self, comm, rank, nranks):
        self.rank, self.nranks = int(rank), int(nranks)
        self._comm = comm            # a comm object, or a factory taking the device Context
        self._made = {}

    def comm_for(self, ctx):
        if not callable(self._comm):
            return self._comm
        if id(ctx) not in self._made:
            self._made[id(ctx)] = self._comm(ctx)
        return self._made[id(ctx)]


_current = None        # set_decomposition() / the launcher's environment
_env_tried = False


def set_decomposition(comm, rank=0, nranks=1):
    """install (comm = None: remove) the x-slab decomposition every Simulation made from here
    on uses.  `comm`: an object with halo_exchange(state, lo, hi) / allreduce_min(x) / gather
    (RcclComm, tests: HostStagedComm) or a factory `comm(ctx)` returning one."""
    global _current, _env_tried
    _current = None if comm is None else Decomposition(comm, rank, nranks)
    _env_tried = comm is not None


def _uid_rendezvous(rank, make_uid, timeout=300.0):
    """rank 0's RCCL unique id to every process of ONE node without torch / MPI: a file under
    /tmp named after the launcher (parent process id, MASTER_PORT, restart count), written
    atomically by rank 0.  PYRO_COMM_ID_FILE names the file for launchers whose workers do not
    share a parent."""
    path = os.environ.get("PYRO_COMM_ID_FILE")
    if not path:
        tag = "_".join(str(x) for x in (os.getppid(), os.environ.get("MASTER_PORT", "0"),
                                        os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"),
                                        os.environ.get("TORCHELASTIC_RUN_ID", "none")))
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pyrohip_uid_{tag}")
    if rank == 0:
        uid = make_uid()
        tmp = f"{path}.{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        import atexit
        atexit.register(lambda: os.path.exists(path) and os.remove(path))
        return uid
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if uid:
                return uid
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise RuntimeError(f"no RCCL unique id from rank 0 after {timeout:.0f} s ({path})")
        time.sleep(0.01)


def _from_environment():
    """one process per GPU under a launcher: RCCL communicator over all WORLD_SIZE ranks on the
    default context (device LOCAL_RANK); None in a single process"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL between processes)
    from . import device
    ndev = device.device_count()
    if ndev < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        # fewer GPUs than ranks: a debugging set-up (the launcher tried on a one-GPU box).  RCCL
        # refuses two ranks on one device of one host, so every rank names itself a host of its
        # own and the communicator runs over the socket transport: same calls, none of the
        # bandwidth.  Never silently.
        if os.environ.get("PYRO_SHARE_GPUS") != "1":
            raise RuntimeError(f"{world} processes but {ndev} GPU(s): one process per GPU is the "
                               "contract (PYRO_SHARE_GPUS=1 lets the ranks share for debugging, "
                               "gpu.decompose=0 runs undecomposed copies)")
        os.environ.setdefault("NCCL_HOSTID", f"pyro2amd-rank{rank}")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        os.environ.setdefault("NCCL_P2P_DISABLE", "1")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    ctx = device.Context.default(device_id=local % max(ndev, 1))
    if ctx.comm_size() == 0:
        uid = _uid_rendezvous(rank, device.Context.comm_unique_id)
        ctx.comm_init(world, rank, uid)
    if ctx.comm_size() != world:
        raise RuntimeError(f"RCCL reports {ctx.comm_size()} ranks, the launcher {world}")
    comm = RcclComm(ctx)
    return Decomposition(lambda c: comm if c is ctx else RcclComm(c), rank, world)


def active_decomposition(rp=None):
    """the Decomposition in force, or None.  gpu.decompose: -1 (default) = decompose when a
    decomposition is installed or the launcher's environment names more than one rank, 0 = never
    (every process runs the whole problem, as the reference would under such a launcher),
    1 = required (an error without ranks)."""
    global _current, _env_tried
    want = -1
    if rp is not None:
        try:
            want = int(rp.get_param("gpu.decompose"))
        except (KeyError, RuntimeError, ValueError, AttributeError):
            want = -1
    if want == 0:
        return None
    if _current is None and not _env_tried:
        _env_tried = True
        _current = _from_environment()
    if _current is None and want == 1:
        raise RuntimeError("gpu.decompose = 1 but there is one process only: start one process "
                           "per GPU (RANK / WORLD_SIZE / LOCAL_RANK, e.g. torch.distributed.run) "
                           "or call pyro2_amd.decomp.set_decomposition()")
    return _current if _current is not None and _current.nranks > 1 else None


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
        # halo sides that are the periodic wrap-around of the global grid (the x ghost
        # rows of the single-domain run), not interior cuts
        self.wrap_lo = bool(periodic and nranks > 1 and rank == 0)
        self.wrap_hi = bool(periodic and nranks > 1 and rank == nranks - 1)

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

    def gather(self, state, dec):
        """COLLECTIVE: the slabs of every rank -> rank 0's host array (qx_global, qy, nvar)
        (x ghost rows of the two end slabs included); None on the other ranks.  Device to
        device over RCCL into a scratch state of the sender's shape, then one download."""
        from . import device
        from ._lib import check, lib
        ng = state.ng
        if dec.rank != 0:
            check(lib().pyrohip_comm_group(1))
            try:
                state.send_rows(0, state.qx, 0)
            finally:
                check(lib().pyrohip_comm_group(0))
            return None
        out = np.empty((dec.nx + 2 * ng, state.qy, state.nvar))
        out[:state.qx] = state.download()
        for r in range(1, dec.nranks):
            tmp = device.DeviceState(self.ctx, dec.counts[r], state.ny, ng,
                                     [["outflow"] * 4] * state.nvar)
            check(lib().pyrohip_comm_group(1))
            try:
                tmp.recv_rows(0, tmp.qx, r)
            finally:
                check(lib().pyrohip_comm_group(0))
            i0 = sum(dec.counts[:r])
            out[i0 + ng:i0 + tmp.qx] = tmp.download()[ng:]
            del tmp
        return out


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

    def __init__(self, ctx, decomp, ny, bcs, params_kw, comm, ng=4, user_bc=None, state=None):
        """state: an existing DeviceState of the slab (boundary table from
        `decomp.comp_var_bcs`, user boundary data set by its owner -- CellCenterData2d behind the
        class surface); params_kw may then be None (the caller passes its own parameter block
        to dt() / evolve()).
        user_bc: (gamma, grav, dy, (ambient rho, u, v, p)) for the hse / ambient
        boundaries on the y sides (compressible/BC.py).  They decompose as they are:
        the halo rows travel whole, WITH the neighbour's y ghost cells, and the y fill
        treats halo rows like interior rows.  One quirk of the reference needs care on
        a grid that is PERIODIC in x: fill_BC_all fills variable after variable, so the
        hse energy of the x ghost rows is built from momenta ghost rows of the previous
        step (`_exchange_fill_wrapped_hse`)."""
        from . import device
        if any(b in ("hse", "ambient") for b in bcs) and user_bc is None and state is None:
            raise ValueError("hse / ambient boundaries need user_bc = (gamma, grav, dy, ambient)")
        self.dec, self.comm = decomp, comm
        if state is None:
            self.state = device.DeviceState(ctx, decomp.nx_local, ny, ng, decomp.comp_var_bcs(bcs))
            if user_bc is not None:
                self.state.set_user_bc(*user_bc)
        else:
            self.state = state
        self._hse_wrap = "hse" in bcs[2:] and (decomp.wrap_lo or decomp.wrap_hi)
        # the artificial viscosity lives on the faces ilo ... ihi of the GLOBAL grid
        # (interface.py:312-364): a slab computes it on its upper face when that face is an
        # interior cut, not when it is the periodic wrap-around (= global face ihi + 1)
        self.avisc_xhi_interior = int(decomp.hi >= 0 and not decomp.wrap_hi)
        if params_kw is not None:
            kw = dict(params_kw)
            kw["avisc_xhi_interior"] = self.avisc_xhi_interior
            self.params = device.make_comp_params(**kw)
        else:
            self.params = None
        # boundary strips first + halo exchange beside the interior strips (kernel_set 2)
        # (not where a wrap-around side meets an hse boundary: that path rewrites halo rows
        # from the host between the steps, _exchange_fill_wrapped_hse)
        self._overlap = bool(getattr(comm, "overlap", False) and (decomp.lo >= 0 or decomp.hi >= 0) and
                             not self._hse_wrap)
        self._rearm = False
        if self._overlap:
            self.state.set_neighbours(decomp.lo, decomp.hi)

    def modified(self):
        """COLLECTIVE: the slab's data is about to be / has been written between two steps
        (upload, upload_rows, a kernel of the caller's).  A step that ran with the overlap on has
        already sent its boundary rows to the neighbours; those rows are stale now, and a rank
        cannot redo the exchange alone (pyrohip_halo_exchange refuses).  Every rank therefore
        drops the posted exchange -- it is let land, then forgotten -- and the next step
        exchanges synchronously; the overlap is switched on again after it."""
        if self._overlap:
            self.state.set_neighbours(-1, -1)
            self._rearm = True

    def upload(self, data):
        """COLLECTIVE (see modified()): replace the slab's data"""
        self.modified()
        self.state.upload(data)

    def upload_rows(self, i0, data):
        """COLLECTIVE (see modified()): replace rows of the slab"""
        self.modified()
        self.state.upload_rows(i0, data)

    def evolve(self, policy, cfl, nsteps, params=None):
        """nsteps of step() enqueued on the device without a host round trip per step
        (pyrohip_comp_evolve: halo exchange, ghost fill, dt policy and update kernels
        back to back; one synchronisation at the end).  Needs the communication inside
        the library (single rank or RcclComm)."""
        if not isinstance(self.comm, (NoComm, RcclComm)):
            raise NotImplementedError("device-side stepping needs RCCL (or a single rank)")
        if self._hse_wrap:
            raise NotImplementedError("hse boundaries on a grid periodic in x step from the host "
                                      "(momenta halo rows of the previous step: step())")
        if isinstance(self.comm, RcclComm) and (self.dec.lo >= 0 or self.dec.hi >= 0):
            self.state.set_neighbours(self.dec.lo, self.dec.hi)
            self._rearm = False
        return self.state.comp_evolve(self.params if params is None else params, cfl, policy, nsteps)

    def _exchange_fill_wrapped_hse(self):
        """halo exchange + ghost fill where a wrap-around side meets an hse boundary.
        The reference fills density, energy, x-momentum, y-momentum one after the other
        (patch.py fill_BC_all), so when the hse energy of an x GHOST row is integrated
        (BC.py:64-84, :95-115: it keeps the kinetic energy of the last interior cell of
        that row) the momenta of that row are still the periodic images of the PREVIOUS
        step.  The rows of the wrap-around halo therefore take their new momenta only
        after density and energy are filled.  (Analytically the kinetic energy cancels
        out of the ghost energy -- e_ghost = E_base + k g rho dy / (gamma - 1) -- so what is
        at stake is the last bit, measured: 1 ulp in the corner cells; bit-identity with
        the single-domain run is the contract here.)  Staged through the host (two extra
        row transfers per step on the two end ranks): an edge case kept exact, not fast."""
        st, ng, nxl = self.state, self.state.ng, self.state.nx
        rows = [r for r, on in ((0, self.dec.wrap_lo), (nxl + ng, self.dec.wrap_hi)) if on]
        old = {r: st.download_rows(r, ng).copy() for r in rows}
        self.comm.halo_exchange(st, self.dec.lo, self.dec.hi)
        new = {}
        for r in rows:
            new[r] = st.download_rows(r, ng).copy()
            mixed = new[r].copy()
            mixed[:, :, 2:4] = old[r][:, :, 2:4]
            st.upload_rows(r, mixed)
        st.fill_bc(0)
        st.fill_bc(1)
        for r in rows:
            cur = st.download_rows(r, ng).copy()
            cur[:, :, 2:4] = new[r][:, :, 2:4]
            st.upload_rows(r, cur)
        st.fill_bc(2)
        st.fill_bc(3)

    def fill(self):
        """COLLECTIVE: CellCenterData2d.fill_BC_all of a slab -- halo rows from the x
        neighbours, then the y / physical ghost fill (the order of array_indexer.py:150-274)"""
        if self._hse_wrap:
            self._exchange_fill_wrapped_hse()
        else:
            self.comm.halo_exchange(self.state, self.dec.lo, self.dec.hi)
            if self._rearm:       # after modified(): that was the synchronous exchange
                self.state.set_neighbours(self.dec.lo, self.dec.hi)
                self._rearm = False
            self.state.fill_bc()

    def dt(self, cfl, params=None):
        """COLLECTIVE: the CFL time step of the GLOBAL state (method_compute_timestep)"""
        params = self.params if params is None else params
        if hasattr(self.comm, "dt_min"):
            return self.comm.dt_min(self.state, params, cfl)
        return self.comm.allreduce_min(self.state.comp_dt(params, cfl))

    def step(self, policy, cfl):
        self.fill()
        dt = policy(self.dt(cfl))
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
