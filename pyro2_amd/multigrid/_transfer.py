"""CellCenterData2d.restrict / prolong (pyro/mesh/patch.py:640-736) through the
multigrid transfer kernels.  Supported where the kernels apply: square 2^k
grids with ng = 1 (the multigrid use case)."""
import numpy as np

from .. import device


def _check(cc):
    g = cc.grid
    if g.ng != 1 or g.nx != g.ny or g.nx & (g.nx - 1):
        raise NotImplementedError("device restrict/prolong needs a square 2^k grid with ng = 1")
    return g


def restrict(cc, varname):
    g = _check(cc)
    if g.nx < 4:
        raise ValueError("grid too small to restrict")
    m = device.DeviceMG(cc.ctx, g.nx, xmin=g.xmin, xmax=g.xmax, ymin=g.ymin, ymax=g.ymax)
    L = m.nlevels - 1
    m.set(L, 2, np.ascontiguousarray(cc.get_var(varname)))
    m.restrict(L)
    cg = g.coarse_like(2)
    out = cg.scratch_array()
    out.v()[:, :] = m.get(L - 1, 1)[1:-1, 1:-1]
    return out


def prolong(cc, varname):
    g = _check(cc)
    m = device.DeviceMG(cc.ctx, 2 * g.nx, xmin=g.xmin, xmax=g.xmax, ymin=g.ymin, ymax=g.ymax)
    L = m.nlevels - 1
    m.set(L - 1, 0, np.ascontiguousarray(cc.get_var(varname)))
    m.zero(L, 0)
    m.prolong_add(L)
    fg = g.fine_like(2)
    out = fg.scratch_array()
    out.v()[:, :] = m.get(L, 0)[1:-1, 1:-1]
    return out
