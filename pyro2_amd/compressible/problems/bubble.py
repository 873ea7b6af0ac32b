"""A hot, under-dense bubble in pressure equilibrium inside an isothermal
hydrostatic atmosphere rises (reference: pyro/compressible/problems/
bubble.py)."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.bubble"
PROBLEM_PARAMS = {"bubble.dens_base": 10.0, "bubble.scale_height": 2.0, "bubble.x_pert": 2.0,
                  "bubble.y_pert": 2.0, "bubble.r_pert": 0.25,
                  "bubble.pert_amplitude_factor": 5.0, "bubble.dens_cutoff": 0.01}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the bubble problem...")
    get = lambda k: rp.get_param("bubble." + k)   # noqa: E731
    gamma, grav = rp.get_param("eos.gamma"), rp.get_param("compressible.grav")
    H, cutoff = get("scale_height"), get("dens_cutoff")
    g = my_data.grid
    # isothermal atmosphere, trapezoidal hydrostatic pressure from the base up
    rho = np.full(g.qy, cutoff)
    p = np.zeros(g.qy)
    for j in range(g.jlo, g.jhi + 1):
        rho[j] = max(get("dens_base") * np.exp(-g.y[j] / H), cutoff)
        p[j] = rho[j] * (H * abs(grav)) if j == g.jlo else \
            p[j - 1] + 0.5 * g.dy * (rho[j] + rho[j - 1]) * grav
    dens, ener = my_data.get_var("density"), my_data.get_var("energy")
    xmom, ymom = my_data.get_var("x-momentum"), my_data.get_var("y-momentum")
    dens[:, :] = rho[np.newaxis, :]
    xmom[:, :] = 0.0
    ymom[:, :] = 0.0
    ener[:, :] = p[np.newaxis, :] / (gamma - 1.0) + 0.5 * (xmom**2 + ymom**2) / dens
    # raise the specific internal energy inside the bubble at constant pressure
    r = np.sqrt((np.asarray(g.x2d) - get("x_pert"))**2 + (np.asarray(g.y2d) - get("y_pert"))**2)
    idx = r <= get("r_pert")
    eint = (ener[idx] - 0.5 * (xmom[idx]**2 - ymom[idx]**2) / dens[idx]) / dens[idx]   # sic
    pres = dens[idx] * eint * (gamma - 1.0)
    eint = eint * get("pert_amplitude_factor")
    dens[idx] = pres / (eint * (gamma - 1.0))
    ener[idx] = dens[idx] * eint + 0.5 * (xmom[idx]**2 + ymom[idx]**2) / dens[idx]


def finalize():
    pass
