"""Gresho vortex: a stationary, rotating vortex whose centrifugal force is
balanced by the pressure gradient; a test of the behaviour at low Mach number
(reference: pyro/compressible/problems/gresho.py)."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.gresho"
PROBLEM_PARAMS = {"gresho.rho0": 1.0, "gresho.r": 0.2, "gresho.mach": 0.1, "gresho.t_r": 1.0}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the Gresho vortex problem...")
    gamma = rp.get_param("eos.gamma")
    rho0, rr = rp.get_param("gresho.rho0"), rp.get_param("gresho.r")
    mach, t_r = rp.get_param("gresho.mach"), rp.get_param("gresho.t_r")
    g = my_data.grid
    x, y = np.asarray(g.x), np.asarray(g.y)
    xc, yc = 0.5 * (x[0] + x[-1]), 0.5 * (y[0] + y[-1])
    q_r = 0.4 * np.pi * (g.xmax - g.xmin) / t_r
    p0 = rho0 * q_r**2 * (5 * rr)**2 / (gamma * mach**2) - 12.5 * rr**2
    x2d, y2d = np.asarray(g.x2d), np.asarray(g.y2d)
    rad = np.sqrt((x2d - xc)**2 + (y2d - yc)**2)
    core, ring = rad < rr, (rad >= rr) & (rad < 2.0 * rr)
    with np.errstate(divide="ignore"):
        lograd = np.log(rad)
    u_phi = np.where(core, 5.0 * rad, np.where(ring, 2.0 - 5.0 * rad, 0.0))
    pres = np.where(core, p0 + 12.5 * rad**2,
                    np.where(ring, p0 + 12.5 * rad**2 +
                             4.0 * (1.0 - 5.0 * rad - np.log(rr) + lograd),
                             p0 + 12.5 * (2.0 * rr)**2 +
                             4.0 * (1.0 - 5.0 * (2.0 * rr) - np.log(rr) + np.log(2.0 * rr))))
    dens = my_data.get_var("density")
    xmom, ymom = my_data.get_var("x-momentum"), my_data.get_var("y-momentum")
    dens[:, :] = rho0
    xmom[:, :] = -dens[:, :] * q_r * u_phi * (y2d - yc) / rad
    ymom[:, :] = dens[:, :] * q_r * u_phi * (x2d - xc) / rad
    my_data.get_var("energy")[:, :] = pres / (gamma - 1.0) + 0.5 * (xmom**2 + ymom**2) / dens
    cs = np.sqrt(gamma * pres / np.asarray(dens))
    print(f"peak Mach number = {np.abs(q_r * u_phi).max() / cs.max()}")


def finalize():
    pass
