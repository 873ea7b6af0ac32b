"""Deterministic initial conditions of the bench-size parity tests:
shared by oracle/gen_fullsize.py (which runs the C oracle on them and stores a lattice of
samples under tests/golden/) and by the GPU tests that compare the device runs with it.
TEST INFRASTRUCTURE."""
import numpy as np

NG = 4


def swe_dam2d_ic(nx, ny=None):
    """a genuinely 2-D dam break for the shallow-water solver: deep water inside an off-centre
    ellipse, shallow outside, plus a smooth ripple and a small sheared flow, on the unit square
    (mesh of inputs.dam.x: outflow in x, reflecting in y): state (qx, qy, 4) in the solver's order
    height, x-momentum, y-momentum, fuel"""
    ny = nx if ny is None else ny
    x = ((np.arange(nx + 2 * NG) - NG + 0.5) / nx)[:, None]
    y = ((np.arange(ny + 2 * NG) - NG + 0.5) / ny)[None, :]
    inside = ((x - 0.43) / 0.21) ** 2 + ((y - 0.55) / 0.14) ** 2 <= 1.0
    h = np.where(inside, 1.0, 0.125) + 0.01 * np.sin(14.0 * x + 3.0 * y) * np.cos(9.0 * y)
    U = np.zeros((nx + 2 * NG, ny + 2 * NG, 4))
    U[..., 0] = h
    U[..., 1] = h * 0.05 * np.sin(6.0 * y)
    U[..., 2] = h * 0.04 * np.cos(5.0 * x)
    U[..., 3] = np.where(inside, 1.0, 0.0) * h
    return U


SWE_BCS = ["outflow", "outflow", "reflect", "reflect"]      # xl xr yl yr (inputs.dam.x + the defaults)


def swe_meta(nx, ny, limiter=1, grav=1.0, cfl=0.8):
    return np.array([nx, ny, NG, 1.0 / nx, 1.0 / ny, grav, limiter, cfl])


def sph_sedov(nx, ny):
    """SphericalPolar Sedov as inputs.sedov.spherical sets it up (r in [0.1, 1], theta in
    [0.785, 2.355], r_init 0.13 from r = 0: the hot cells are those with r < 0.13): returns the
    grid object, the geometry arrays, the state and the boundary list"""
    from pyro2_amd.mesh import patch
    gamma = 1.4
    grid = patch.SphericalPolar(nx, ny, ng=NG, xmin=0.1, xmax=1.0, ymin=0.785, ymax=2.355)
    U0 = np.zeros((grid.qx, grid.qy, 4))
    U0[:, :, 0] = 1.0
    U0[:, :, 1] = 1.e-6 / (gamma - 1.0)
    U0[:, :, 1][np.asarray(grid.x2d) < 0.13] = 1.e6
    # a weak angular modulation, so that the run is not one-dimensional
    U0[:, :, 0] *= 1.0 + 0.02 * np.cos(3.0 * np.asarray(grid.y2d))
    bcs = ["reflect-odd", "outflow", "outflow", "outflow"]
    return grid, grid.device_geometry(), U0, bcs


def lattice(I, n=64):
    """what a fixture keeps of a full-size interior: an n x n lattice of cells, row / column sums
    per variable, the per-variable maxima"""
    si, sj = max(1, I.shape[0] // n), max(1, I.shape[1] // n)
    return {"samples": I[::si, ::sj].copy(), "row_sums": I.sum(axis=1), "col_sums": I.sum(axis=0),
            "umax": np.abs(I).max(axis=(0, 1))}


def assert_lattice(I, g, tol, floor=None, what=""):
    """device interior I against a fixture written by lattice(): element-wise on the samples
    (|a - b| <= tol (|b| + floor_n), floor_n = 1e-3 x the variable's maximum unless given), sums
    to tol x (their own scale)"""
    n = g["samples"].shape[0]
    si, sj = max(1, I.shape[0] // n), max(1, I.shape[1] // n)
    S = I[::si, ::sj]
    umax = g["umax"]
    for v in range(I.shape[2]):
        fl = (1e-3 * umax[v] if floor is None else floor[v]) + 1e-300
        err = np.abs(S[..., v] - g["samples"][..., v]) / (np.abs(g["samples"][..., v]) + fl)
        assert err.max() <= tol, (what, "lattice", v, float(err.max()))
        for ax, key in ((1, "row_sums"), (0, "col_sums")):
            ref = g[key][:, v]
            scale = max(np.abs(ref).max(), I.shape[ax] * umax[v] * 1e-3)
            assert np.abs(I[..., v].sum(axis=ax) - ref).max() <= tol * scale, (what, key, v)
