"""Grid2d / Cartesian2d / CellCenterData2d with the call surface of
pyro/mesh/patch.py:42-794, cell data resident on the GPU.

Differences in mechanism (not in behaviour):
  * 2-d coordinate arrays (x2d, y2d, ..., Lx, Ly, Ax, Ay, V) are built lazily
    on first access -- the reference builds 11 full (qx,qy) arrays eagerly
    (patch.py:137-147, 210-233), which does not fit host memory at 16384^2.
  * CellCenterData2d keeps a planar copy of `data` on the device.  Host and
    device copies carry validity flags: touching `.data` / `get_var()`
    downloads if the device is newer and (because writes through NumPy views
    cannot be observed) marks the device copy stale; the next device
    operation uploads first.  In Pyro.run_sim's steady loop nothing touches
    the host copy, so the state never leaves HBM.
"""
import sys

import numpy as np

from .. import device
from .._lib import BC_CODE
from ..util import msg
from . import boundary as bnd
from .array_indexer import ArrayIndexer


class Grid2d:
    """the discretisation: nx x ny zones plus ng ghost cells per side on
    [xmin,xmax] x [ymin,ymax]  (patch.py:42-189)"""

    coord_type = None

    def __init__(self, nx, ny, *, ng=1, xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0, slab=None):
        """slab: a decomp.SlabDecomp -- this rank's x-slab of the nx x ny grid (one process per
        GPU).  The object then describes rows [i0, i0 + nx_local) of the global
        grid: `nx`, `qx`, `ilo`, `ihi` and every array are the SLAB's, `xmin` / `xmax` / `dx`
        stay the global ones and the coordinates are those of the global rows, bit for bit (the
        global row number goes through the reference's expression), so a problem's init_data
        fills a slab exactly as it fills those rows of the whole grid.  `nx_global` / `i0` tell
        the two apart (nx / 0 on an undecomposed grid)."""
        self.slab = slab
        self.nx_global = int(nx)
        self.i0 = 0 if slab is None else int(slab.i0)
        if slab is not None:
            if slab.nx != int(nx):
                raise ValueError("the slab decomposition belongs to another grid")
            nx_here = slab.nx_local
        else:
            nx_here = int(nx)
        self.nx, self.ny, self.ng = int(nx_here), int(ny), int(ng)
        self.qx, self.qy = int(2 * ng + self.nx), int(2 * ng + ny)
        self.xmin, self.xmax, self.ymin, self.ymax = xmin, xmax, ymin, ymax
        self.ilo, self.ihi = self.ng, self.ng + self.nx - 1
        self.jlo, self.jhi = self.ng, self.ng + self.ny - 1
        self.ic = self.ilo + self.nx // 2 - 1
        self.jc = self.jlo + self.ny // 2 - 1
        # 1-d coordinates, same expressions as patch.py:121-134 (on the global row numbers)
        self.dx = (xmax - xmin) / self.nx_global
        self.xl = (np.arange(self.i0, self.i0 + self.qx) - ng) * self.dx + xmin
        self.xr = (np.arange(self.i0, self.i0 + self.qx) + 1.0 - ng) * self.dx + xmin
        self.x = 0.5 * (self.xl + self.xr)
        self.dy = (ymax - ymin) / ny
        self.yl = (np.arange(self.qy) - ng) * self.dy + ymin
        self.yr = (np.arange(self.qy) + 1.0 - ng) * self.dy + ymin
        self.y = 0.5 * (self.yl + self.yr)
        self._lazy = {}

    # ---- lazily built 2-d coordinate arrays ----------------------------
    def _mesh(self, key, a, b, which):
        if key not in self._lazy:
            aa, bb = np.meshgrid(a, b, indexing="ij")
            self._lazy[key] = ArrayIndexer(d=aa if which == 0 else bb, grid=self)
        return self._lazy[key]

    x2d = property(lambda self: self._mesh("x2d", self.x, self.y, 0))
    y2d = property(lambda self: self._mesh("y2d", self.x, self.y, 1))
    xl2d = property(lambda self: self._mesh("xl2d", self.xl, self.yl, 0))
    yl2d = property(lambda self: self._mesh("yl2d", self.xl, self.yl, 1))
    xr2d = property(lambda self: self._mesh("xr2d", self.xr, self.yr, 0))
    yr2d = property(lambda self: self._mesh("yr2d", self.xr, self.yr, 1))

    def scratch_array(self, *, nvar=1, dtype=np.float64):
        shape = (self.qx, self.qy) if nvar == 1 else (self.qx, self.qy, nvar)
        return ArrayIndexer(d=np.zeros(shape, dtype=dtype), grid=self)

    def coarse_like(self, N):
        return Grid2d(self.nx // N, self.ny // N, ng=self.ng, xmin=self.xmin,
                      xmax=self.xmax, ymin=self.ymin, ymax=self.ymax)

    def fine_like(self, N):
        return Grid2d(self.nx * N, self.ny * N, ng=self.ng, xmin=self.xmin,
                      xmax=self.xmax, ymin=self.ymin, ymax=self.ymax)

    def __str__(self):
        return f"2-d grid: nx = {self.nx}, ny = {self.ny}, ng = {self.ng}"

    def __eq__(self, other):
        keys = ("nx", "ny", "ng", "xmin", "xmax", "ymin", "ymax")
        return all(getattr(self, k) == getattr(other, k) for k in keys) and \
            (getattr(self, "i0", 0), getattr(self, "nx_global", self.nx)) == \
            (getattr(other, "i0", 0), getattr(other, "nx_global", getattr(other, "nx", None)))

    __hash__ = None


class Cartesian2d(Grid2d):
    """Cartesian geometry: Lx = dx, Ly = dy, Ax = Ly, Ay = Lx, V = dx dy
    (patch.py:192-239); the constant 2-d arrays are built on demand"""

    def __init__(self, nx, ny, *, ng=1, xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0, slab=None):
        super().__init__(nx, ny, ng=ng, xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax, slab=slab)
        self.coord_type = 0

    def _const(self, key, value):
        if key not in self._lazy:
            self._lazy[key] = ArrayIndexer(np.full((self.qx, self.qy), value), grid=self)
        return self._lazy[key]

    Lx = property(lambda self: self._const("Lx", self.dx))
    Ly = property(lambda self: self._const("Ly", self.dy))
    Ax = property(lambda self: self.Ly)
    Ay = property(lambda self: self.Lx)
    dlogAx = property(lambda self: self._const("dlogAx", 0.0))
    dlogAy = property(lambda self: self._const("dlogAy", 0.0))
    V = property(lambda self: self._const("V", self.dx * self.dy))

    def __str__(self):
        return (f"Cartesian 2D Grid: xmin = {self.xmin}, xmax = {self.xmax}, "
                f"ymin = {self.ymin}, ymax = {self.ymax}, "
                f"nx = {self.nx}, ny = {self.ny}, ng = {self.ng}")


class SphericalPolar(Grid2d):
    """Spherical polar grid with azimuthal symmetry, x = r and y = theta
    (patch.py:242-318): cell sizes, face areas, volumes and the dlog(area)
    factors as 2-d arrays, built with the reference's expressions.  The
    compressible solver hands them to the device once
    (pyrohip_state_set_geometry); `device_geometry()` also carries the sines
    that artificial_viscosity evaluates (compressible/interface.py:345-347)."""

    def __init__(self, nx, ny, *, ng=1, xmin=0.2, xmax=1.0, ymin=0.0, ymax=1.0):
        super().__init__(nx, ny, ng=ng, xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax)
        assert ymin >= 0.0 and ymax <= np.pi, "y or \u03b8 should be within [0, \u03c0]."
        assert xmin - ng * self.dx >= 0.0, \
            "xmin (r-direction), must be large enough so ghost cell doesn't have negative x."
        self.coord_type = 1
        self.Lx = ArrayIndexer(np.full((self.qx, self.qy), self.dx), grid=self)   # dr
        self.Ly = ArrayIndexer(self.x2d * self.dy, grid=self)                     # r dtheta
        # |r_{i-1/2}^2 2 pi (cos(theta_{j+1/2}) - cos(theta_{j-1/2}))|
        self.Ax = np.abs(-2.0 * np.pi * self.xl2d**2 * (np.cos(self.yr2d) - np.cos(self.yl2d)))
        # |pi sin(theta_{j-1/2}) (r_{i+1/2}^2 - r_{i-1/2}^2)|
        self.Ay = np.abs(np.pi * np.sin(self.yl2d) * (self.xr2d**2 - self.xl2d**2))
        self.dlogAx = 2.0 / self.x2d
        self.dlogAy = 1.0 / (np.tan(self.y2d) * self.x2d)
        self.V = np.abs(-2.0 * np.pi / 3.0 * (np.cos(self.yr2d) - np.cos(self.yl2d)) *
                        (self.xr2d - self.xl2d) *
                        (self.xr2d**2 + self.xl2d**2 + self.xr2d * self.xl2d))

    def device_geometry(self):
        j = np.arange(self.qy)
        out = {n: np.ascontiguousarray(getattr(self, n)) for n in
               ("Lx", "Ly", "Ax", "Ay", "V", "dlogAx", "dlogAy", "x2d")}
        # scalar by scalar like the reference's loop body
        out["sint"] = np.array([np.sin((jj + 0.5 - self.ng) * self.dy + self.ymin) for jj in j])
        out["sinb"] = np.array([np.sin((jj - 0.5 - self.ng) * self.dy + self.ymin) for jj in j])
        out["sinc"] = np.array([np.sin((jj - self.ng) * self.dy + self.ymin) for jj in j])
        # The 2-d arrays are products of a row factor and a column factor, in this order of
        # operations (the expressions of __init__ above, patch.py:262-312): handed over as 1-d
        # tables, the kernels rebuild Ax = |A B|, Ay = |C D|, V = |(E F) G|, dlogAy = 1 / (T x)
        # in registers with the SAME bits (checked below) instead of reading eight planes
        xl, xr, x = self.xl, self.xr, self.x
        A = (-2.0 * np.pi) * xl**2
        D = xr**2 - xl**2
        F = xr - xl
        G = xr**2 + xl**2 + xr * xl
        B = np.cos(self.yr) - np.cos(self.yl)
        Cc = np.pi * np.sin(self.yl)
        E = (-2.0 * np.pi / 3.0) * B
        T = np.tan(self.y)
        rowf = np.array([A, D, F, G, x * self.dy, 2.0 / x, x])
        colf = np.array([B, Cc, E, T])
        with np.errstate(divide="ignore", invalid="ignore"):
            same = np.array_equal(out["Ax"], np.abs(A[:, None] * B[None, :])) and \
                np.array_equal(out["Ay"], np.abs(Cc[None, :] * D[:, None])) and \
                np.array_equal(out["V"], np.abs((E[None, :] * F[:, None]) * G[:, None])) and \
                np.array_equal(out["dlogAy"], 1.0 / (T[None, :] * x[:, None]), equal_nan=True) and \
                np.array_equal(out["Ly"], np.broadcast_to(rowf[4][:, None], out["Ly"].shape)) and \
                np.array_equal(out["dlogAx"], np.broadcast_to(rowf[5][:, None], out["Ly"].shape)) and \
                np.all(out["Lx"] == self.dx) and np.all(np.isfinite(out["dlogAy"])) and \
                np.array_equal(out["x2d"], np.broadcast_to(x[:, None], out["Ly"].shape))
        if same:       # (else: the kernels keep reading the planes)
            out["rowf"], out["colf"] = rowf, colf
        return out

    def coarse_like(self, N):
        raise NotImplementedError("multigrid hierarchies are Cartesian")

    fine_like = coarse_like

    def __str__(self):
        return ("Spherical Polar 2D Grid: Define x : r, y : \u03b8. "
                f"xmin (r) = {self.xmin}, xmax= {self.xmax}, "
                f"ymin = {self.ymin}, ymax = {self.ymax}, "
                f"nx = {self.nx}, ny = {self.ny}, ng = {self.ng}")


def _bc_row(bc, device_user_bc=False):
    """BC codes of one variable; user types without a device kernel get 0
    (their callback overwrites the ghost cells afterwards)"""
    return [bnd.device_bcs[b] if (device_user_bc and b in bnd.device_bcs)
            else BC_CODE.get(b, 0) if b not in bnd.ext_bcs else 0
            for b in bc.sides()]


def _fill_host_array(ctx, arr, n, bc):
    """ArrayIndexer.fill_ghost for a stand-alone host array: one round trip
    through the device ghost-fill kernel"""
    g = arr.g
    plane = np.ascontiguousarray(arr if arr.ndim == 2 else arr[:, :, n], dtype=np.float64)
    st = device.DeviceState(ctx, g.nx, g.ny, g.ng, [_bc_row(bc)])
    st.upload_var(0, plane)
    st.fill_bc(0)
    out = st.download_var(0)
    _apply_inhomogeneous(out, g, bc)
    if arr.ndim == 2:
        arr[:, :] = out
    else:
        arr[:, :, n] = out


def _apply_inhomogeneous(a, g, bc):
    """first-ghost-cell values for inhomogeneous Dirichlet / Neumann data
    (array_indexer.py:166-183,196-215); host side, O(perimeter)"""
    xl, xr, yl, yr = bc.values()
    if xl is not None:
        a[g.ilo - 1, :] = a[g.ilo, :] - g.dx * xl if bc.xlb in ("outflow", "neumann") \
            else 2 * xl - a[g.ilo, :]
    if xr is not None:
        a[g.ihi + 1, :] = a[g.ihi, :] + g.dx * xr if bc.xrb in ("outflow", "neumann") \
            else 2 * xr - a[g.ihi, :]
    if yl is not None:
        a[:, g.jlo - 1] = a[:, g.jlo] - g.dy * yl if bc.ylb in ("outflow", "neumann") \
            else 2 * yl - a[:, g.jlo]
    if yr is not None:
        a[:, g.jhi + 1] = a[:, g.jhi] + g.dy * yr if bc.yrb in ("outflow", "neumann") \
            else 2 * yr - a[:, g.jhi]


class CellCenterData2d:
    """cell-centred state on a grid: register_var()* -> create() -> use
    (patch.py:315-794).  Storage order of `data` is (qx, qy, nvar)."""

    def __init__(self, grid, *, dtype=np.float64, ctx=None):
        # the kernels compute in float64 (the reference's default); any other dtype
        # (patch.py:315 takes one; its own unit tests use int) is kept on the host in
        # that type and converted on the way to / from the device
        self.grid = grid
        self.dtype = np.dtype(dtype)
        self.names = []
        self.vars = self.names   # same alias as the reference
        self.nvar = 0
        self.ivars = []
        self.aux = {}
        self.derives = []
        self.BCs = {}
        self.t = -1.0
        self.initialized = 0
        self._ctx = ctx
        self._host = None
        self._dev = None
        self._host_valid = True
        self._dev_valid = False
        # deferred ghost fill (fill_BC_all with lazy_fill, see there)
        self.lazy_fill = False
        self._fill_pending = False
        # NumPy views handed out by get_var() / .data keep the host copy's
        # ArrayIndexer alive through their .base chain, so its reference count
        # tells whether user code still holds one (see device_modified)
        self._root = None
        self._root_refs = 0
        self._warned_views = False
        # x-slab of a decomposed run (grid.slab: decomp.SlabDecomp): halo rows from the x
        # neighbours replace the x ghost fill on the cut faces; every operation marked
        # COLLECTIVE below must then be called by all ranks (pyro's driver does: every process
        # runs the same script)
        self.slab = getattr(grid, "slab", None)
        self._comm = None
        self._slab_driver = None      # compressible: decomp.SlabCompressible around the device state

    # ---- construction ---------------------------------------------------
    def register_var(self, name, bc):
        if self.initialized == 1:
            msg.fail("ERROR: grid already initialized")
        self.names.append(name)
        self.nvar += 1
        self.BCs[name] = bc

    def set_aux(self, keyword, value):
        self.aux[keyword] = value

    def get_aux(self, keyword):
        return self.aux.get(keyword)

    def add_derived(self, func):
        self.derives.append(func)

    def add_ivars(self, ivars):
        self.ivars = ivars

    def create(self):
        if self.initialized == 1:
            msg.fail("ERROR: grid already initialized")
        g = self.grid
        self._set_host(np.zeros((g.qx, g.qy, self.nvar), dtype=self.dtype))
        self._host_valid, self._dev_valid = True, False
        self.initialized = 1

    def _set_host(self, arr):
        self._root = np.ascontiguousarray(arr, dtype=self.dtype)
        self._host = ArrayIndexer(self._root, grid=self.grid)
        self._root_refs = sys.getrefcount(self._host)

    def _views_alive(self):
        """does user code hold a NumPy view of the host copy (a get_var() result
        kept across steps, like `dens = sim.cc_data.get_var("density")` in the
        reference's scripts)?"""
        return self._host is not None and sys.getrefcount(self._host) > self._root_refs

    # ---- host / device coherence ---------------------------------------
    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = device.Context.default()
        return self._ctx

    def device_state(self, fuse_fill=False):
        """the DeviceState with a current copy of the data (uploads if the host
        copy was touched since the last device operation).  A ghost fill that
        fill_BC_all() deferred (see there) is carried out now unless the caller
        takes it over (fuse_fill=True, then take_pending_fill())"""
        if self._fill_pending and not fuse_fill:
            self._fill_pending = False
            self._fill_now()
        if self._dev is None:
            g = self.grid
            dub = self._device_user_bc()
            rows = [_bc_row(self.BCs[n], dub) for n in self.names]
            if self.slab is not None:       # cut faces: halo rows, not a boundary rule
                rows = self.slab.var_bcs(rows)
            self._dev = device.DeviceState(self.ctx, g.nx, g.ny, g.ng, rows)
        if not self._dev_valid:
            if self._slab_driver is not None:
                # (COLLECTIVE) rows a step has already sent to the neighbours are stale now
                self._slab_driver.modified()
            self._dev.upload(np.asarray(self._host, dtype=np.float64))
            self._dev_valid = True
        return self._dev

    @property
    def comm(self):
        """what moves halo rows of a decomposed run (decomp.RcclComm; tests: a gloo transport)"""
        if self._comm is None and self.slab is not None:
            from .. import decomp
            dec = decomp.active_decomposition()
            if dec is None:
                raise RuntimeError("a slab grid without a decomposition in force "
                                   "(pyro2_amd.decomp.set_decomposition)")
            self._comm = dec.comm_for(self.ctx)
        return self._comm

    def _download_to_host(self):
        """device copy -> the host array (same memory: views stay valid)"""
        host = np.asarray(self._host)
        if host.dtype == np.float64:
            self._dev.download(host)
        else:
            host[...] = self._dev.download()

    def device_modified(self):
        """to be called after a kernel changed the device copy.  In the
        reference the arrays are updated in place, so a view taken before a
        step shows the new state after it and can be written through at any
        time.  While user code holds such a view the host copy is therefore
        refreshed right away (same memory, the view sees it) and uploaded again
        before the next kernel (the view may have been written): correct, at
        the price of two PCIe transfers per step -- scripts that re-fetch
        get_var() after each step (or never look) run without them."""
        self._dev_valid = True
        self._host_valid = False
        if self._views_alive():
            if not self._warned_views:
                self._warned_views = True
                msg.warning("a view of the simulation data (get_var / .data) is held across "
                            "device steps: keeping the host copy coherent costs a download and "
                            "an upload per step")
            self._download_to_host()
            self._host_valid = True
            self._dev_valid = False

    def take_pending_fill(self):
        """True if fill_BC_all() deferred a ghost fill that the caller now does
        as part of its own kernel"""
        p, self._fill_pending = self._fill_pending, False
        return p

    def _host_rw(self):
        if self._fill_pending:           # the host copy must see filled ghost cells
            self._fill_pending = False
            self._fill_now()
        if not self._host_valid:
            self._download_to_host()
            self._host_valid = True
        # views handed out may be written through: the device copy is stale
        self._dev_valid = False
        return self._host

    @property
    def data(self):
        return self._host_rw()

    @data.setter
    def data(self, value):
        self._set_host(np.array(value, dtype=self.dtype))
        self._host_valid, self._dev_valid = True, False
        self._fill_pending = False

    # ---- variable access ------------------------------------------------
    def get_var(self, name):
        """stored variable: writable view into `data`; otherwise the first
        derived-variable function that knows the name (patch.py:476-508)"""
        if name in self.names:
            return self.get_var_by_index(self.names.index(name))
        for f in self.derives:
            try:
                var = f(self, name)
            except TypeError:
                var = f(self, name, self.ivars, self.grid)
            if len(var) > 0:
                return var
        raise KeyError(f"name {name} is not valid") from None

    def get_var_readonly(self, name):
        """like get_var for callers that only READ (host-side diagnostics such
        as the tracer particles): the device copy stays current"""
        keep = self._dev_valid
        try:
            return self.get_var(name)
        finally:
            self._dev_valid = keep

    def get_var_by_index(self, n):
        return ArrayIndexer(d=self._host_rw()[:, :, n], grid=self.grid)

    def get_vars(self):
        return ArrayIndexer(d=self._host_rw(), grid=self.grid)

    def zero(self, name):
        self._host_rw()[:, :, self.names.index(name)] = 0.0

    def min(self, name, *, ng=0):
        """(COLLECTIVE on a slab: the minimum over the whole grid)"""
        n = self.names.index(name)
        if self._dev_valid and self._dev is not None:
            m = self.device_state().minmax(n, buf=ng)[0]
        else:
            m = np.min(self._host.v(buf=ng, n=n))
        return m if self.slab is None else self.comm.allreduce_min(float(m))

    def max(self, name, *, ng=0):
        n = self.names.index(name)
        if self._dev_valid and self._dev is not None:
            m = self.device_state().minmax(n, buf=ng)[1]
        else:
            m = np.max(self._host.v(buf=ng, n=n))
        return m if self.slab is None else -self.comm.allreduce_min(-float(m))

    # ---- boundary conditions -------------------------------------------
    def _device_user_bc(self):
        """the hse / ambient / ramp boundaries run on the device for the
        compressible state (4 variables in pyro's order; hse and ambient on the
        y sides only, ramp anywhere but the upper x side)"""
        used = {b for n in self.names for b in self.BCs[n].sides() if b in bnd.device_bcs}
        if used and used <= set(bnd.const_bcs):
            # constant-value ghost cells ("moving_lid"): any state, upper y side only
            return all(b not in bnd.const_bcs for n in self.names
                       for b in self.BCs[n].sides()[:3])
        if self.names != ["density", "energy", "x-momentum", "y-momentum"]:
            return False
        for n in self.names:
            xl, xr = self.BCs[n].sides()[:2]
            if xr in bnd.device_bcs or (xl in bnd.device_bcs and xl != "ramp"):
                return False
        return True

    def _has_host_bc(self, name):
        bc = self.BCs[name]
        dev_ok = self._device_user_bc()
        return any(b in bnd.ext_bcs and not (dev_ok and b in bnd.device_bcs)
                   for b in bc.sides()) or \
            any(v is not None for v in bc.values())

    def _push_user_bc(self, st):
        """hand gamma, grav and the ambient state (aux data, set by the solver
        and the problem setup) to the device ghost fill"""
        used = {b for n in self.names for b in self.BCs[n].sides() if b in bnd.device_bcs}
        if not used:
            return
        if used & set(bnd.const_bcs) and not getattr(st, "_const_bc_pushed", False):
            for n, name in enumerate(self.names):
                b = self.BCs[name].yrb
                if b in bnd.const_bcs:
                    st.set_const_bc(n, bnd.const_bcs[b](name))
            st._const_bc_pushed = True
        if used & {"hse", "ambient"}:
            amb = [self.aux.get(k, 0.0) for k in
                   ("ambient_rho", "ambient_u", "ambient_v", "ambient_p")]
            st.set_user_bc(self.get_aux("gamma"), self.get_aux("grav"), self.grid.dy, amb)
        if "ramp" in used:      # time dependent: the shock front moves with t
            from ..compressible import BC as comp_bc
            st.set_ramp_bc(**comp_bc.ramp_params(self.grid, self.get_aux("gamma"), self.t))

    def fill_BC_all(self):
        """ghost fill of every variable.  A solver whose step kernel can do the
        fill itself (advection: index remap at load, one launch instead of four)
        sets `lazy_fill`: the fill is then only NOTED here and carried out by the
        next thing that touches the data -- the solver's kernel (fused), or any
        other device / host access (a real fill, device_state / _host_rw)."""
        if self.lazy_fill and self._lazy_fill_ok():
            self._fill_pending = True
            return
        self._fill_now()

    def _lazy_fill_ok(self):
        simple = ("outflow", "reflect-even", "reflect-odd", "periodic")
        # (a slab's fill starts with the halo exchange, a collective: never deferred)
        return self.slab is None and self._dev_valid and self._dev is not None and \
            not any(self._has_host_bc(n) for n in self.names) and \
            all(b in simple for n in self.names for b in self.BCs[n].sides())

    def _fill_now(self):
        if self.slab is not None:
            # COLLECTIVE: halo rows from the x neighbours first, then the y / physical sides
            # (the x ghost rows are filled before the y ghost columns, array_indexer.py:150-274:
            # a halo row arrives with the neighbour's y ghost cells of the previous fill and
            # the y fill that follows rewrites them like those of an interior row)
            if any(self._has_host_bc(n) for n in self.names):
                msg.fail("ERROR: host-side boundary callbacks / inhomogeneous boundary values "
                         "are not carried by a decomposed run")
            st = self.device_state()
            self._push_user_bc(st)
            if self._slab_driver is not None:
                self._slab_driver.fill()
            else:
                self.comm.halo_exchange(st, self.slab.lo, self.slab.hi)
                st.fill_bc(-1)
            self.device_modified()
            return
        if not any(self._has_host_bc(n) for n in self.names):
            st = self.device_state()
            self._push_user_bc(st)
            st.fill_bc(-1)     # one launch pair for all variables
            self.device_modified()
            return
        for name in self.names:
            self.fill_BC(name)

    def fill_BC(self, name):
        """device ghost fill (ArrayIndexer.fill_ghost semantics), then any
        user-defined boundary callbacks on the host copy (patch.py:582-624)"""
        n = self.names.index(name)
        bc = self.BCs[name]
        if self.slab is not None:
            # COLLECTIVE; the halo rows of a slab travel for all variables at once
            self._fill_now()
            return
        st = self.device_state()
        self._push_user_bc(st)
        st.fill_bc(n)
        self.device_modified()
        if not self._has_host_bc(name):
            return
        host = self._host_rw()
        _apply_inhomogeneous(host[:, :, n], self.grid, bc)
        for side, tag in zip(bc.sides(), ("xlb", "xrb", "ylb", "yrb")):
            if side in bnd.ext_bcs:
                try:
                    bnd.ext_bcs[side](side, tag, name, self, self.ivars)
                except TypeError:
                    bnd.ext_bcs[side](side, tag, name, self)

    # ---- grid transfer (patch.py:640-736) ------------------------------
    def restrict(self, varname, N=2):
        """average onto a grid coarser by N = 2 or 4 (multigrid/_transfer.py)"""
        from ..multigrid import _transfer
        return _transfer.restrict(self, varname, N)

    def prolong(self, varname):
        from ..multigrid import _transfer
        return _transfer.prolong(self, varname)

    # ---- output ---------------------------------------------------------
    def write(self, filename):
        """pyro's HDF5 layout (through h5py, or util/h5pure.py when h5py is
        not installed: util/h5lite.py picks)"""
        from ..util import h5lite
        with h5lite.open_file(filename, "w") as f:
            self.write_data(f)

    def gather(self):
        """COLLECTIVE on a slab: the whole grid's CellCenterData2d on rank 0 (host copy; grid,
        variables, boundary objects, aux data and time of this object), None on the other ranks;
        the object itself on an undecomposed grid."""
        if self.slab is None:
            return self
        if self._fill_pending:
            self._fill_pending = False
            self._fill_now()
        full = self.comm.gather(self.device_state(), self.slab)
        if full is None:
            return None
        g = self.grid
        whole = type(g)(g.nx_global, g.ny, ng=g.ng, xmin=g.xmin, xmax=g.xmax, ymin=g.ymin,
                        ymax=g.ymax)
        out = CellCenterData2d(whole, dtype=self.dtype, ctx=self._ctx)
        for name in self.names:
            out.register_var(name, self.BCs[name])
        out.aux = dict(self.aux)
        out.derives = list(self.derives)
        out.ivars = self.ivars
        out.create()
        out._set_host(full)
        out._host_valid, out._dev_valid = True, False
        out.t = self.t
        return out

    def write_data(self, f):
        """HDF5 layout of patch.py:750-788: groups aux / grid / state/<var>"""
        if self.slab is not None:
            raise RuntimeError("a slab writes through gather(): rank 0 holds the whole grid "
                               "(Simulation.write does this)")
        gaux = f.create_group("aux")
        for k, v in self.aux.items():
            gaux.attrs[k] = v
        g = self.grid
        ggrid = f.create_group("grid")
        for k in ("nx", "ny", "ng", "xmin", "xmax", "ymin", "ymax"):
            ggrid.attrs[k] = getattr(g, k)
        if g.coord_type is not None:
            ggrid.attrs["coord_type"] = g.coord_type
        gstate = f.create_group("state")
        for n, name in enumerate(self.names):
            gvar = gstate.create_group(name)
            gvar.create_dataset("data", data=self.get_var_by_index(n).v())
            bc = self.BCs[name]
            for tag, val in zip(("xlb", "xrb", "ylb", "yrb"), bc.sides()):
                gvar.attrs[tag] = val

    def pretty_print(self, var, fmt=None):
        self.get_var(var).pretty_print(fmt=fmt)

    def __str__(self):
        if self.initialized == 0:
            return "CellCenterData2d object not yet initialized"
        g = self.grid
        s = f"cc data: nx = {g.nx}, ny = {g.ny}, ng = {g.ng}\n"
        s += f"         nvars = {self.nvar}\n         variables:\n"
        for name in self.names:
            bc = self.BCs[name]
            s += f"{name:>16s}: min: {self.min(name):15.10f}    max: {self.max(name):15.10f}\n"
            s += f"{' ':>16s}  BCs: -x: {bc.xlb:12s} +x: {bc.xrb:12s} -y: {bc.ylb:12s} +y: {bc.yrb:12s}\n"
        return s


def cell_center_data_clone(old):
    """a new CellCenterData2d with the same grid, variables, BCs and aux data
    and a copy of the data (patch.py:951-980)"""
    new = CellCenterData2d(old.grid, dtype=old.dtype, ctx=old._ctx)
    new._comm = old._comm
    for name in old.names:
        new.register_var(name, old.BCs[name])
    new.aux = dict(old.aux)
    new.derives = list(old.derives)
    new.create()
    new.data[:, :, :] = np.asarray(old.data)
    return new
