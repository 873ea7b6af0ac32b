"""Edge-centred coefficients of a cell-centred field and their restriction
(pyro/multigrid/edge_coeffs.py:1-54): the container the reference's variable-coefficient
solvers build their operator from.  Host-side NumPy: the device solvers derive the same
quantities themselves (`k_vc_edges`, `k_vc_edges_restrict` in csrc/multigrid.hip); this
module keeps code that imports `pyro.multigrid.edge_coeffs` working."""
import numpy as np


class EdgeCoeffs:
    """x[i, j] = eta_{i-1/2, j} / dx^2 and y[i, j] = eta_{i, j-1/2} / dy^2 on the faces of the
    interior cells (lower faces ilo ... ihi + 1, jlo ... jhi + 1), zero elsewhere"""

    def __init__(self, g, eta, empty=False):
        self.grid = g
        if empty:
            return
        e = np.asarray(eta)
        ii = slice(g.ilo, g.ihi + 2)
        jj = slice(g.jlo, g.jhi + 2)
        self.x = g.scratch_array()
        self.y = g.scratch_array()
        self.x[ii, jj] = 0.5 * (e[g.ilo - 1:g.ihi + 1, jj] + e[ii, jj])
        self.y[ii, jj] = 0.5 * (e[ii, g.jlo - 1:g.jhi + 1] + e[ii, jj])
        self.x /= g.dx**2
        self.y /= g.dy**2

    def restrict(self):
        """the coefficients of the grid coarser by 2: an x face of a coarse cell is the
        average of the two fine x faces it covers (same i, rows 2J and 2J + 1), likewise y;
        renormalised to the coarse spacing"""
        fg = self.grid
        cg = fg.coarse_like(2)
        c = EdgeCoeffs(cg, None, empty=True)
        cx, cy = cg.scratch_array(), cg.scratch_array()
        fx, fy = np.asarray(self.x), np.asarray(self.y)
        # coarse faces ilo ... ihi + 1 sit on the fine faces ilo, ilo + 2, ...
        fi = slice(fg.ilo, fg.ihi + 3, 2)
        fj = slice(fg.jlo, fg.jhi + 1, 2)
        fj1 = slice(fg.jlo + 1, fg.jhi + 2, 2)
        cx[cg.ilo:cg.ihi + 2, cg.jlo:cg.jhi + 1] = 0.5 * (fx[fi, fj] + fx[fi, fj1])
        fi = slice(fg.ilo, fg.ihi + 1, 2)
        fi1 = slice(fg.ilo + 1, fg.ihi + 2, 2)
        fj = slice(fg.jlo, fg.jhi + 3, 2)
        cy[cg.ilo:cg.ihi + 1, cg.jlo:cg.jhi + 2] = 0.5 * (fy[fi, fj] + fy[fi1, fj])
        c.x = cx * fg.dx**2 / cg.dx**2
        c.y = cy * fg.dy**2 / cg.dy**2
        return c
