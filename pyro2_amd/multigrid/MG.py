"""MG.CellCenterMG2d with the call surface of pyro/multigrid/MG.py:85-778.

The level hierarchy (v, f, r on 2^2 ... nx^2 grids, ng = 1) lives on the
device; smoothing, residual, restriction, prolongation, norms and the V-cycle
recursion are HIP kernels (csrc/multigrid.hip).  Host arrays cross the PCIe
bus only in init_solution / init_RHS / get_solution*.

Callers in pyro (diffusion, incompressible, ...) construct a new object every
time step; construction here is cheap: no per-level coordinate
arrays are built until somebody asks for x2d / y2d.
"""
import numpy as np

from .. import device
from ..mesh import boundary as bnd
from ..mesh import patch
from ..mesh.array_indexer import ArrayIndexer
from ..util import msg


class _LevelData:
    """what callers see in `mg.grids[level]`: get_var('v'|'f'|'r') returns a
    host copy of the level array; `grid` is the level's Grid2d"""

    _VAR = {"v": 0, "f": 1, "r": 2}

    def __init__(self, mg, level, grid, bc_v, bc):
        self._mg, self._level, self.grid = mg, level, grid
        self.names = ["v", "f", "r"]
        self.BCs = {"v": bc_v, "f": bc, "r": bc}

    def get_var(self, name):
        return ArrayIndexer(self._mg._dev.get(self._level, self._VAR[name]), grid=self.grid)

    def set_var(self, name, data):
        self._mg._dev.set(self._level, self._VAR[name], np.asarray(data))

    def zero(self, name):
        self._mg._dev.zero(self._level, self._VAR[name])

    def fill_BC(self, name):
        self._mg._dev.fill_bc(self._level, self._VAR[name])


class CellCenterMG2d:
    def __init__(self, nx, ny, ng=1, xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0,
                 xl_BC_type="dirichlet", xr_BC_type="dirichlet",
                 yl_BC_type="dirichlet", yr_BC_type="dirichlet",
                 xl_BC=None, xr_BC=None, yl_BC=None, yr_BC=None,
                 alpha=0.0, beta=-1.0, nsmooth=10, nsmooth_bottom=50, verbose=0,
                 aux_field=None, aux_bc=None, true_function=None, vis=0, vis_title="",
                 ctx=None, slab=None):
        if nx != ny:
            raise ValueError("ERROR: multigrid currently requires nx = ny")
        if (xmax - xmin) != (ymax - ymin):
            raise ValueError("ERROR: multigrid currently requires a square domain")
        if ng != 1:
            raise ValueError("the device multigrid uses ng = 1 (the reference default)")
        if aux_field is not None:
            raise NotImplementedError("aux fields belong to the variable-coefficient "
                                      "subclasses (SURVEY.md 8 row f1)")
        self.nx, self.ny, self.ng = nx, ny, ng
        self.xmin, self.xmax, self.ymin, self.ymax = xmin, xmax, ymin, ymax
        self.alpha, self.beta = alpha, beta
        self.nsmooth, self.nsmooth_bottom = nsmooth, nsmooth_bottom
        self.max_cycles = 100
        self.verbose = verbose
        if true_function is not None:
            self.true_function = true_function
        self.small = 1.e-16
        self.initialized_rhs = 0
        self.vis, self.vis_title, self.frame = vis, vis_title, 0

        self.ctx = ctx if ctx is not None else device.Context.default()
        types = (xl_BC_type, xr_BC_type, yl_BC_type, yr_BC_type)
        self._dev = device.DeviceMG(self.ctx, nx, xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax,
                                    bcs=types, alpha=alpha, beta=beta, nsmooth=nsmooth,
                                    nsmooth_bottom=nsmooth_bottom)
        self.nlevels = self._dev.nlevels
        # x-slab decomposition over the GPUs of a node (one process per GPU):
        # slab = (comm, rank, nranks[, collapse_n]) or what SlabMG.set_decomposition()
        # installed; every rank constructs the same solver and calls the same methods
        from .slab import SlabMG
        self._slab = None
        slab = slab if slab is not None else SlabMG._default
        if slab is not None and type(self) is CellCenterMG2d:
            comm, rank, nranks = slab[:3]
            collapse = slab[3] if len(slab) > 3 else 256
            if nranks > 1 and nx > collapse:
                self._slab = SlabMG(self._dev, comm, rank, nranks, collapse_n=collapse,
                                    nsmooth=nsmooth)
        bc = bnd.BC(xlb=types[0], xrb=types[1], ylb=types[2], yrb=types[3])
        self.grids = []
        n = 2
        for lev in range(self.nlevels):
            g = patch.Grid2d(n, n, ng=1, xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax)
            bc_v = bc
            if lev == self.nlevels - 1:
                # inhomogeneous data only on the finest level (MG.py:231-245)
                bc_v = bnd.BC(xlb=types[0], xrb=types[1], ylb=types[2], yrb=types[3],
                              xl_func=xl_BC, xr_func=xr_BC, yl_func=yl_BC, yr_func=yr_BC,
                              grid=g)
                for side, vals in enumerate(bc_v.values()):
                    if vals is not None:
                        self._dev.set_bcval(side, np.asarray(vals, dtype=np.float64))
            self.grids.append(_LevelData(self, lev, g, bc_v, bc))
            n *= 2
        sg = self.grids[-1].grid
        self.soln_grid = sg
        self.ilo, self.ihi, self.jlo, self.jhi = sg.ilo, sg.ihi, sg.jlo, sg.jhi
        self.x, self.y, self.dx, self.dy = sg.x, sg.y, sg.dx, sg.dy
        self.source_norm = 0.0
        self.num_cycles = 0
        self.residual_error = 1.e33
        self.relative_error = 1.e33
        self.current_cycle = -1
        self.current_level = -1
        self.up_or_down = ""

    # lazily built like Grid2d's
    x2d = property(lambda self: self.soln_grid.x2d)
    y2d = property(lambda self: self.soln_grid.y2d)

    def grid_info(self, level, indent=0):
        g = self.grids[level].grid
        print(f"{indent * ' '}level: {level}, grid: {g.nx} x {g.ny}")

    # ---- data in / out ---------------------------------------------------
    def init_solution(self, data):
        self._dev.set(self.nlevels - 1, 0, np.array(data, dtype=np.float64))

    def init_zeros(self):
        self._dev.zero(self.nlevels - 1, 0)

    def init_RHS(self, data):
        self._dev.set(self.nlevels - 1, 1, np.array(data, dtype=np.float64))
        self.source_norm = self._dev.init_rhs_norm()
        if self.verbose:
            print("Source norm = ", self.source_norm)
        self.initialized_rhs = 1

    def get_solution(self, grid=None):
        v = self.grids[-1].get_var("v")
        if grid is None:
            return v.copy()
        myg = self.soln_grid
        assert grid.dx == myg.dx and grid.dy == myg.dy
        sol = grid.scratch_array()
        sol.v(buf=1)[:, :] = v.v(buf=1)
        return sol

    def get_solution_gradient(self, grid=None):
        """centred difference of the solution (MG.py:447-481); a host-side
        convenience on the downloaded solution"""
        myg = self.soln_grid
        og = myg if grid is None else grid
        assert og.dx == myg.dx and og.dy == myg.dy
        v = self.grids[-1].get_var("v")
        gx, gy = og.scratch_array(), og.scratch_array()
        gx.v()[:, :] = 0.5 * (v.ip(1) - v.ip(-1)) / myg.dx
        gy.v()[:, :] = 0.5 * (v.jp(1) - v.jp(-1)) / myg.dy
        return gx, gy

    def get_solution_object(self):
        return self.grids[-1]

    # ---- the numerics (all on the device) --------------------------------
    def _compute_residual(self, level):
        self._dev.residual(level)

    def smooth(self, level, nsmooth):
        self._dev.smooth(level, nsmooth)
        # the reference's last fill_BC also sets the corner ghosts
        self._dev.fill_bc(level, 0)

    def v_cycle(self, level):
        if self._slab is not None and level == self.nlevels - 1:
            self._slab.vcycle()
            self._slab.gather_solution()
            return
        self._dev.vcycle(level)

    def solve(self, rtol=1.e-11):
        if not self.initialized_rhs:
            msg.fail("ERROR: RHS not initialized")
        if self.verbose:
            print("source norm = ", self.source_norm)
        if self._slab is not None:
            nc, res, rel = self._slab.solve(rtol=rtol, max_cycles=self.max_cycles)
        else:
            nc, res, rel = self._dev.solve(rtol=rtol, max_cycles=self.max_cycles)
        self.num_cycles = nc
        self.residual_error = res
        self.relative_error = rel
        if self.verbose:
            print(f"{nc} V-cycles: relative err = {rel}, residual err = {res}\n")
