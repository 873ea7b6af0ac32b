"""Kelvin-Helmholtz instability with two smoothed shear layers at y = 1/4 and
y = 3/4 (McNally et al. 2012 set-up) and an optional vertical bulk velocity.
Reference: pyro/compressible/problems/kh.py."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.kh"
PROBLEM_PARAMS = {"kh.rho_1": 1.0, "kh.u_1": -1.0, "kh.rho_2": 2.0,
                  "kh.u_2": 1.0, "kh.bulk_velocity": 0.0}

_WIDTH = 0.025    # smoothing length of the layers
_W0 = 0.01        # amplitude of the seed perturbation
_PRES = 2.5


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the Kelvin-Helmholtz problem...")
    rho_1, u_1, rho_2, u_2 = (rp.get_param("kh." + k) for k in ("rho_1", "u_1", "rho_2", "u_2"))
    bulk = rp.get_param("kh.bulk_velocity")
    gamma = rp.get_param("eos.gamma")
    g = my_data.grid
    x2d, y2d = np.asarray(g.x2d), np.asarray(g.y2d)
    dens = my_data.get_var("density")
    xmom = my_data.get_var("x-momentum")
    ymom = my_data.get_var("y-momentum")
    ener = my_data.get_var("energy")

    vm, rhom = 0.5 * (u_1 - u_2), 0.5 * (rho_1 - rho_2)
    # four horizontal bands: (mask, outer state, sign of the blend, exponent)
    bands = ((y2d < 0.25, (rho_1, u_1), -1.0, (y2d - 0.25) / _WIDTH),
             ((y2d >= 0.25) & (y2d < 0.5), (rho_2, u_2), 1.0, (0.25 - y2d) / _WIDTH),
             ((y2d >= 0.5) & (y2d < 0.75), (rho_2, u_2), 1.0, (y2d - 0.75) / _WIDTH),
             (y2d >= 0.75, (rho_1, u_1), -1.0, (0.75 - y2d) / _WIDTH))
    vel = np.zeros_like(x2d)
    for mask, (rho_b, u_b), sgn, expo in bands:
        blend = np.exp(expo[mask])
        dens[mask] = rho_b + sgn * rhom * blend
        vel[mask] = u_b + sgn * vm * blend
    xmom[:, :] = vel * dens
    ymom[:, :] = dens * (bulk + _W0 * np.sin(4 * np.pi * x2d))
    ener[:, :] = _PRES / (gamma - 1.0) + 0.5 * (xmom**2 + ymom**2) / dens


def finalize():
    pass
