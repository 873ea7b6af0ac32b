"""Isothermal atmosphere in hydrostatic equilibrium; it should stay static
(a test of the gravity source term and the hse boundary).  Reference:
pyro/compressible/problems/hse.py."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.hse"
PROBLEM_PARAMS = {"hse.dens0": 1.0, "hse.h": 1.0}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the HSE problem...")
    gamma = rp.get_param("eos.gamma")
    grav = rp.get_param("compressible.grav")
    dens0, H = rp.get_param("hse.dens0"), rp.get_param("hse.h")
    print("dens0 = ", dens0)
    cs2 = H * abs(grav)            # isothermal sound speed squared
    g = my_data.grid
    y = np.asarray(g.y)
    # exponential density on the interior rows; the pressure is the discrete
    # (trapezoidal) hydrostatic integral upwards from the bottom row
    rho_y = np.zeros(g.qy)
    rho_y[g.jlo:g.jhi + 1] = dens0 * np.exp(-y[g.jlo:g.jhi + 1] / H)
    p_y = np.zeros(g.qy)
    p_y[g.jlo] = rho_y[g.jlo] * cs2
    for j in range(g.jlo + 1, g.jhi + 1):
        p_y[j] = p_y[j - 1] + 0.5 * g.dy * (rho_y[j] + rho_y[j - 1]) * grav
    dens = my_data.get_var("density")
    xmom = my_data.get_var("x-momentum")
    ymom = my_data.get_var("y-momentum")
    ener = my_data.get_var("energy")
    dens[:, :] = rho_y[np.newaxis, :]
    xmom[:, :] = 0.0
    ymom[:, :] = 0.0
    with np.errstate(invalid="ignore", divide="ignore"):   # 0/0 in the y ghost rows
        ener[:, :] = p_y[np.newaxis, :] / (gamma - 1.0) + 0.5 * (xmom**2 + ymom**2) / dens


def finalize():
    pass
