"""Single-mode Rayleigh-Taylor instability: heavy fluid (dens2) on top of light
fluid (dens1) in hydrostatic balance, with a cosine velocity perturbation
localised at the interface (reference: pyro/compressible/problems/rt.py)."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.rt"
PROBLEM_PARAMS = {"rt.dens1": 1.0, "rt.dens2": 2.0, "rt.amp": 1.0,
                  "rt.sigma": 0.1, "rt.p0": 10.0}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the rt problem...")
    gamma = rp.get_param("eos.gamma")
    grav = rp.get_param("compressible.grav")
    dens1, dens2 = rp.get_param("rt.dens1"), rp.get_param("rt.dens2")
    p0, amp, sigma = (rp.get_param("rt." + k) for k in ("p0", "amp", "sigma"))
    g = my_data.grid
    dens = my_data.get_var("density")
    xmom = my_data.get_var("x-momentum")
    ymom = my_data.get_var("y-momentum")
    ener = my_data.get_var("energy")

    # stratification on the interior rows (all columns); ghost rows stay 0
    # until the first boundary fill
    ymid = 0.5 * (g.ymin + g.ymax)
    y = np.asarray(g.y)
    inside = np.zeros(g.qy, dtype=bool)
    inside[g.jlo:g.jhi + 1] = True
    lower = inside & (y < ymid)
    upper = inside & ~(y < ymid)
    rho_y = np.where(lower, dens1, np.where(upper, dens2, 0.0))
    p_y = np.where(lower, p0 + dens1 * grav * y,
                   np.where(upper, p0 + dens1 * grav * ymid + dens2 * grav * (y - ymid), 0.0))
    dens[:, :] = rho_y[np.newaxis, :]

    L = g.xmax - g.xmin
    x2d, y2d = np.asarray(g.x2d), np.asarray(g.y2d)
    mode = np.cos(2.0 * np.pi * x2d / L) + np.cos(2.0 * np.pi * (L - x2d) / L)
    xmom[:, :] = 0.0
    ymom[:, :] = amp * 0.5 * mode * np.exp(-(y2d - ymid)**2 / sigma**2)
    ymom *= dens
    with np.errstate(invalid="ignore", divide="ignore"):   # 0/0 in the y ghost rows
        ener[:, :] = p_y[np.newaxis, :] / (gamma - 1.0) + 0.5 * (xmom**2 + ymom**2) / dens


def finalize():
    pass
