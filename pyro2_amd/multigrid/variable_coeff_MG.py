"""VarCoeffCCMG2d: multigrid for div(eta grad phi) = f with the call surface
of pyro/multigrid/variable_coeff_MG.py:23-213.  The cell-centred coefficient
is uploaded once; edge coefficients of every level, the coefficient-aware
smoother and residual live on the device (csrc/multigrid.hip, k_vc_*)."""
import numpy as np

from ..mesh.array_indexer import ArrayIndexer
from . import MG


class _EdgeCoeffsView:
    """edge_coeffs[level].x / .y as host copies (edge_coeffs.py:1-54)"""

    def __init__(self, mg, level):
        self._mg, self._level = mg, level
        self.grid = mg.grids[level].grid

    @property
    def x(self):
        return ArrayIndexer(self._mg._dev.get(self._level, 4), grid=self.grid)

    @property
    def y(self):
        return ArrayIndexer(self._mg._dev.get(self._level, 5), grid=self.grid)


class VarCoeffCCMG2d(MG.CellCenterMG2d):
    def __init__(self, nx, ny, xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0,
                 xl_BC_type="dirichlet", xr_BC_type="dirichlet",
                 yl_BC_type="dirichlet", yr_BC_type="dirichlet",
                 nsmooth=10, nsmooth_bottom=50, verbose=0, coeffs=None, coeffs_bc=None,
                 true_function=None, vis=0, vis_title="", ctx=None):
        super().__init__(nx, ny, ng=1, xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax,
                         xl_BC_type=xl_BC_type, xr_BC_type=xr_BC_type,
                         yl_BC_type=yl_BC_type, yr_BC_type=yr_BC_type,
                         alpha=0.0, beta=0.0, nsmooth=nsmooth, nsmooth_bottom=nsmooth_bottom,
                         verbose=verbose, true_function=true_function, vis=vis,
                         vis_title=vis_title, ctx=ctx)
        c = np.asarray(coeffs)
        if c.shape != (nx + 2, ny + 2):
            raise IndexError("coefficient array not the same size as multigrid problem")
        self._dev.set_coeffs(c, coeffs_bc.sides())
        self.edge_coeffs = [_EdgeCoeffsView(self, lev) for lev in range(self.nlevels)]
        for lev, gl in enumerate(self.grids):
            gl.names = ["v", "f", "r", "coeffs"]
            gl._VAR = dict(gl._VAR, coeffs=3)
