"""GeneralMG2d: multigrid for
    alpha phi + div(beta grad phi) + gamma . grad phi = f
with the call surface of pyro/multigrid/general_MG.py:22-242.  `coeffs` is a
CellCenterData2d with the fields alpha, beta, gamma_x, gamma_y on the fine
grid; they are uploaded once, restricted down the hierarchy and (beta) averaged
to the edges on the device; the smoother and the residual are the GEN variants
of the variable-coefficient kernels (csrc/multigrid.hip, k_vc_*<true>)."""
import numpy as np

from ..mesh.array_indexer import ArrayIndexer
from . import MG

_FIELDS = ("alpha", "beta", "gamma_x", "gamma_y")


class _BetaEdgeView:
    """beta_edge[level].x / .y as host copies (edge_coeffs.py:1-54)"""

    def __init__(self, mg, level):
        self._mg, self._level = mg, level
        self.grid = mg.grids[level].grid

    @property
    def x(self):
        return ArrayIndexer(self._mg._dev.get(self._level, 4), grid=self.grid)

    @property
    def y(self):
        return ArrayIndexer(self._mg._dev.get(self._level, 5), grid=self.grid)


class GeneralMG2d(MG.CellCenterMG2d):
    def __init__(self, nx, ny, xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0,
                 xl_BC_type="dirichlet", xr_BC_type="dirichlet",
                 yl_BC_type="dirichlet", yr_BC_type="dirichlet",
                 xl_BC=None, xr_BC=None, yl_BC=None, yr_BC=None,
                 nsmooth=10, nsmooth_bottom=50, verbose=0, coeffs=None,
                 true_function=None, vis=0, vis_title="", ctx=None):
        super().__init__(nx, ny, ng=1, xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax,
                         xl_BC_type=xl_BC_type, xr_BC_type=xr_BC_type,
                         yl_BC_type=yl_BC_type, yr_BC_type=yr_BC_type,
                         xl_BC=xl_BC, xr_BC=xr_BC, yl_BC=yl_BC, yr_BC=yr_BC,
                         alpha=0.0, beta=0.0, nsmooth=nsmooth, nsmooth_bottom=nsmooth_bottom,
                         verbose=verbose, true_function=true_function, vis=vis,
                         vis_title=vis_title, ctx=ctx)
        arrs = [np.asarray(coeffs.get_var(n)) for n in _FIELDS]
        for a in arrs:
            if a.shape != (nx + 2, ny + 2):
                raise IndexError("coefficient array not the same size as multigrid problem")
        self._dev.set_general_coeffs(*arrs, [coeffs.BCs[n].sides() for n in _FIELDS])
        self.beta_edge = [_BetaEdgeView(self, lev) for lev in range(self.nlevels)]
        for gl in self.grids:
            gl.names = ["v", "f", "r", "beta", "alpha", "gamma_x", "gamma_y"]
            gl._VAR = dict(gl._VAR, beta=3, alpha=6, gamma_x=7, gamma_y=8)
