"""A localised heat source near the bottom of an adiabatically stratified
atmosphere drives a buoyant plume (reference: pyro/compressible/problems/
plume.py).  Source on the device through `heating_profile`."""
import numpy as np

from ...util import msg
from ._atmosphere import adiabatic_density

DEFAULT_INPUTS = "inputs.plume"
PROBLEM_PARAMS = {"plume.dens_base": 10.0, "plume.scale_height": 4.0, "plume.x_pert": 2.0,
                  "plume.y_pert": 2.0, "plume.r_pert": 0.25, "plume.e_rate": 0.1,
                  "plume.dens_cutoff": 0.01}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the plume problem...")
    gamma = rp.get_param("eos.gamma")
    grav = rp.get_param("compressible.grav")
    H, rho0 = rp.get_param("plume.scale_height"), rp.get_param("plume.dens_base")
    g = my_data.grid
    rho = adiabatic_density(g, gamma, rho0, H, rp.get_param("plume.dens_cutoff"))
    # hydrostatic pressure: trapezoidal integration upwards from the base
    p = np.zeros(g.qy)
    p[g.jlo] = H * rho0 * abs(grav)
    for j in range(g.jlo + 1, g.jhi + 1):
        p[j] = p[j - 1] + 0.5 * g.dy * (rho[j] + rho[j - 1]) * grav
    dens = my_data.get_var("density")
    xmom, ymom = my_data.get_var("x-momentum"), my_data.get_var("y-momentum")
    dens[:, :] = rho[np.newaxis, :]
    xmom[:, :] = 0.0
    ymom[:, :] = 0.0
    my_data.get_var("energy")[:, :] = p[np.newaxis, :] / (gamma - 1.0) + \
        0.5 * (xmom**2 + ymom**2) / dens


def heating_profile(myg, rp):
    dist = np.sqrt((np.asarray(myg.x2d) - rp.get_param("plume.x_pert"))**2 +
                   (np.asarray(myg.y2d) - rp.get_param("plume.y_pert"))**2)
    return rp.get_param("plume.e_rate"), np.exp(-(dist / rp.get_param("plume.r_pert"))**2)


def source_terms(myg, U, ivars, rp):
    rate, prof = heating_profile(myg, rp)
    S = myg.scratch_array(nvar=ivars.nvar)
    S[:, :, ivars.iener] = U[:, :, ivars.idens] * rate * prof
    return S


def finalize():
    pass
