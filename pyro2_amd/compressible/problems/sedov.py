"""Sedov blast wave: uniform gas with the explosion energy deposited inside
r_init around the domain centre, sub-sampled nsub x nsub per zone (the
reference's sedov problem, pyro/compressible/problems/sedov.py:15-93).

`sedov_state` builds the conserved array (optionally a slab of rows only, for
grids that do not fit host memory in one piece); it is vectorised over the
candidate zones but every number goes through the same arithmetic as the
reference's per-zone loop, so the array is bit-identical (tests/golden)."""
import math

import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.sedov"
PROBLEM_PARAMS = {"sedov.r_init": 0.1,   # radius of the initial perturbation
                  "sedov.nsub": 4}       # sub-samples per zone and direction


def sedov_state(nx, ny, ng, xmin, xmax, ymin, ymax, gamma, r_init, nsub, i0=0, ni=None):
    """(ni, qy, 4) block of rows [i0, i0+ni) of the (qx, qy, 4) initial state,
    component order density, energy, x-momentum, y-momentum"""
    qx, qy = nx + 2 * ng, ny + 2 * ng
    ni = qx if ni is None else ni
    dx, dy = (xmax - xmin) / nx, (ymax - ymin) / ny
    ii = np.arange(i0, i0 + ni)
    xl = (ii - ng) * dx + xmin
    x = 0.5 * (xl + ((ii + 1.0 - ng) * dx + xmin))
    jj = np.arange(qy)
    yl = (jj - ng) * dy + ymin
    y = 0.5 * (yl + ((jj + 1.0 - ng) * dy + ymin))
    xctr, yctr = 0.5 * (xmin + xmax), 0.5 * (ymin + ymax)
    E_sedov = 1.0
    U = np.zeros((ni, qy, 4))
    U[:, :, 0] = 1.0
    U[:, :, 1] = 1.e-5 / (gamma - 1.0)
    ci = np.nonzero(np.abs(x - xctr) < 2.0 * r_init + dx)[0]
    cj = np.nonzero(np.abs(y - yctr) < 2.0 * r_init + dy)[0]
    if len(ci) == 0 or len(cj) == 0:
        return U
    X, Y = np.meshgrid(x[ci], y[cj], indexing="ij")
    near = np.nonzero(np.sqrt((X - xctr)**2 + (Y - yctr)**2) < 2.0 * r_init)
    si, sj = ci[near[0]], cj[near[1]]
    sub = np.arange(nsub) + 0.5
    xs = xl[si][:, None] + (dx / nsub) * sub[None, :]
    ys = yl[sj][:, None] + (dy / nsub) * sub[None, :]
    dist = np.sqrt((xs[:, :, None] - xctr)**2 + (ys[:, None, :] - yctr)**2)
    n_in = np.count_nonzero(dist <= r_init, axis=(1, 2))
    p = n_in * (gamma - 1.0) * E_sedov / (math.pi * r_init * r_init) + \
        (nsub * nsub - n_in) * 1.e-5
    p = p / (nsub * nsub)
    U[si, sj, 1] = p / (gamma - 1.0)
    return U


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the sedov problem...")
    g = my_data.grid
    if g.coord_type != 0:
        # SphericalPolar branch (sedov.py:84-93): a hot sphere r < r_init
        gamma = rp.get_param("eos.gamma")
        my_data.get_var("density")[:, :] = 1.0
        my_data.get_var("x-momentum")[:, :] = 0.0
        my_data.get_var("y-momentum")[:, :] = 0.0
        ener = my_data.get_var("energy")
        ener[:, :] = 1.e-6 / (gamma - 1.0)
        ener[np.asarray(g.x2d) < rp.get_param("sedov.r_init")] = 1.e6
        return
    # (an x-slab of a decomposed run holds rows [i0, i0 + qx) of the nx_global-row grid)
    U = sedov_state(g.nx_global, g.ny, g.ng, rp.get_param("mesh.xmin"), rp.get_param("mesh.xmax"),
                    rp.get_param("mesh.ymin"), rp.get_param("mesh.ymax"),
                    rp.get_param("eos.gamma"), rp.get_param("sedov.r_init"),
                    rp.get_param("sedov.nsub"), i0=g.i0, ni=g.qx)
    for n, name in enumerate(("density", "energy", "x-momentum", "y-momentum")):
        my_data.get_var(name)[:, :] = U[:, :, n]


def finalize():
    print("""
          Radially averaged profiles of this run can be compared with the
          exact cylindrical Sedov solution (analysis/sedov_compare.py in pyro).
          """)
