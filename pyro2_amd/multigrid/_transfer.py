"""CellCenterData2d.restrict / prolong (pyro/mesh/patch.py:640-736).  On the grids the
multigrid solvers use (square 2^k, ng = 1) they run through the multigrid transfer
kernels; the solvers themselves never come here (their V-cycle restricts and prolongs
inside csrc/multigrid.hip).  Any other grid -- rectangular, other ghost widths,
restriction by 4: the general utility of the reference's mesh class -- is a few NumPy
slice operations on the host, as in the reference."""
import numpy as np

from .. import device


def _device_shape(g):
    return g.ng == 1 and g.nx == g.ny and g.nx >= 4 and not (g.nx & (g.nx - 1))


def _restrict_host(cc, varname, N):
    """average of the N x N fine cells under a coarse cell, summed in the order of
    patch.py:657-673"""
    g = cc.grid
    f = np.asarray(cc.get_var(varname))
    cg = g.coarse_like(N)
    out = cg.scratch_array()

    def child(di, dj):
        return f[g.ilo + di:g.ihi + 1:N, g.jlo + dj:g.jhi + 1:N]
    acc = None
    for dj in range(N):
        for di in range(N):
            acc = child(di, dj) if acc is None else acc + child(di, dj)
    out.v()[:, :] = 0.25 * acc if N == 2 else acc / 16.0
    return out


def _prolong_host(cc, varname):
    """piecewise-linear reconstruction with centred slopes, averaged over the four
    children (patch.py:678-736)"""
    g = cc.grid
    c = np.asarray(cc.get_var(varname))
    ii, jj = slice(g.ilo, g.ihi + 1), slice(g.jlo, g.jhi + 1)
    c0 = c[ii, jj]
    m_x = 0.5 * (c[g.ilo + 1:g.ihi + 2, jj] - c[g.ilo - 1:g.ihi, jj])
    m_y = 0.5 * (c[ii, g.jlo + 1:g.jhi + 2] - c[ii, g.jlo - 1:g.jhi])
    fg = g.fine_like(2)
    out = fg.scratch_array()
    o = np.asarray(out)
    a, b = fg.ilo, fg.jlo
    o[a:fg.ihi + 1:2, b:fg.jhi + 1:2] = c0 - 0.25 * m_x - 0.25 * m_y
    o[a + 1:fg.ihi + 1:2, b:fg.jhi + 1:2] = c0 + 0.25 * m_x - 0.25 * m_y
    o[a:fg.ihi + 1:2, b + 1:fg.jhi + 1:2] = c0 - 0.25 * m_x + 0.25 * m_y
    o[a + 1:fg.ihi + 1:2, b + 1:fg.jhi + 1:2] = c0 + 0.25 * m_x + 0.25 * m_y
    return out


def restrict(cc, varname, N=2):
    g = cc.grid
    if N not in (2, 4):
        raise ValueError("restriction is only allowed by 2 or 4")
    if N != 2 or not _device_shape(g):
        return _restrict_host(cc, varname, N)
    m = device.DeviceMG(cc.ctx, g.nx, xmin=g.xmin, xmax=g.xmax, ymin=g.ymin, ymax=g.ymax)
    L = m.nlevels - 1
    m.set(L, 2, np.ascontiguousarray(cc.get_var(varname)))
    m.restrict(L)
    cg = g.coarse_like(2)
    out = cg.scratch_array()
    out.v()[:, :] = m.get(L - 1, 1)[1:-1, 1:-1]
    return out


def prolong(cc, varname):
    g = cc.grid
    if not _device_shape(g):
        return _prolong_host(cc, varname)
    m = device.DeviceMG(cc.ctx, 2 * g.nx, xmin=g.xmin, xmax=g.xmax, ymin=g.ymin, ymax=g.ymax)
    L = m.nlevels - 1
    m.set(L - 1, 0, np.ascontiguousarray(cc.get_var(varname)))
    m.zero(L, 0)
    m.prolong_add(L)
    fg = g.fine_like(2)
    out = fg.scratch_array()
    out.v()[:, :] = m.get(L, 0)[1:-1, 1:-1]
    return out
