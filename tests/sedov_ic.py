"""Sedov initial condition for tests / bench (host side, NumPy).

Same recipe as the reference problem (pyro/compressible/problems/sedov.py:
15-93, inputs.sedov): rho = 1, p = 1e-5, explosion energy 1 spread over the
cells inside r_init with nsub^2 sub-sampling.  Vectorised over the candidate
cells; every element goes through the same arithmetic as the reference loop,
so the array is bit-identical (checked against tests/golden in
test_host_api.py).
"""
import math

import numpy as np


def sedov_ic(nx, ny=None, ng=4, r_init=0.01, nsub=4, gamma=1.4,
             xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0, i0=0, ni=None):
    """returns (U (rows,qy,4), meta, bcs).  i0/ni select a slab of rows of the
    full (qx, qy) array so that huge grids can be generated piecewise."""
    ny = nx if ny is None else ny
    qx, qy = nx + 2 * ng, ny + 2 * ng
    ni = qx if ni is None else ni
    dx, dy = (xmax - xmin) / nx, (ymax - ymin) / ny
    ii = np.arange(i0, i0 + ni)
    xl = (ii - ng) * dx + xmin
    xr = (ii + 1.0 - ng) * dx + xmin
    x = 0.5 * (xl + xr)
    jj = np.arange(qy)
    yl = (jj - ng) * dy + ymin
    yr = (jj + 1.0 - ng) * dy + ymin
    y = 0.5 * (yl + yr)
    xctr, yctr = 0.5 * (xmin + xmax), 0.5 * (ymin + ymax)
    U = np.zeros((ni, qy, 4))
    U[:, :, 0] = 1.0
    p0 = 1.e-5
    U[:, :, 1] = p0 / (gamma - 1.0)
    # candidate cells: dist < 2 r_init
    ci = np.nonzero(np.abs(x - xctr) < 2.0 * r_init + dx)[0]
    cj = np.nonzero(np.abs(y - yctr) < 2.0 * r_init + dy)[0]
    if len(ci) and len(cj):
        X, Y = np.meshgrid(x[ci], y[cj], indexing="ij")
        dist = np.sqrt((X - xctr) ** 2 + (Y - yctr) ** 2)
        sel = np.nonzero(dist < 2.0 * r_init)
        si, sj = ci[sel[0]], cj[sel[1]]
        sub = np.arange(nsub) + 0.5
        xs = xl[si][:, None] + (dx / nsub) * sub[None, :]
        ys = yl[sj][:, None] + (dy / nsub) * sub[None, :]
        d = np.sqrt((xs[:, :, None] - xctr) ** 2 + (ys[:, None, :] - yctr) ** 2)
        n_in = np.count_nonzero(d <= r_init, axis=(1, 2))
        p = n_in * (gamma - 1.0) * 1.0 / (math.pi * r_init * r_init) + \
            (nsub * nsub - n_in) * 1.e-5
        p = p / (nsub * nsub)
        U[si, sj, 1] = p / (gamma - 1.0)
    meta = np.array([nx, ny, ng, dx, dy, gamma, 2, 1, 0.75, 0.85, 0.33, 0.1, 0.0, 0.8])
    return U, meta, ["outflow"] * 4
