"""A heated layer at some height above the bottom drives convection in an
adiabatically stratified atmosphere; reflecting bottom, `ambient` top, sponge
in the low-density region (reference: pyro/compressible/problems/
convection.py).  Source on the device through `heating_profile`."""
import numpy as np

from ...util import msg
from ._atmosphere import adiabatic_density

DEFAULT_INPUTS = "inputs.convection"
PROBLEM_PARAMS = {"convection.dens_base": 10.0, "convection.scale_height": 4.0,
                  "convection.y_height": 2.0, "convection.thickness": 0.25,
                  "convection.e_rate": 0.1, "convection.dens_cutoff": 0.01}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the convection problem...")
    gamma = rp.get_param("eos.gamma")
    grav = rp.get_param("compressible.grav")
    H, rho0 = rp.get_param("convection.scale_height"), rp.get_param("convection.dens_base")
    cutoff = rp.get_param("convection.dens_cutoff")
    g = my_data.grid
    rho = adiabatic_density(g, gamma, rho0, H, cutoff)
    # adiabat p = p_base (rho / rho_base)^gamma; constant above the atmosphere
    pres_base = H * rho0 * abs(grav)
    p = np.zeros(g.qy)
    for j in range(g.jlo, g.jhi + 1):
        if j == g.jlo:
            p[j] = pres_base
        elif rho[j] <= cutoff + 1.e-30:
            p[j] = p[j - 1]
        else:
            p[j] = pres_base * (rho[j] / rho0)**gamma
    my_data.set_aux("ambient_rho", cutoff)
    my_data.set_aux("ambient_u", 0.0)
    my_data.set_aux("ambient_v", 0.0)
    my_data.set_aux("ambient_p", p[g.jlo:g.jhi + 1].min())
    dens = my_data.get_var("density")
    xmom, ymom = my_data.get_var("x-momentum"), my_data.get_var("y-momentum")
    dens[:, :] = rho[np.newaxis, :]
    xmom[:, :] = 0.0
    ymom[:, :] = 0.0
    p2d = np.broadcast_to(p[np.newaxis, :], (g.qx, g.qy))
    my_data.get_var("energy")[:, :] = p2d / (gamma - 1.0)
    # seed: random velocities of up to 5 % of the sound speed inside the atmosphere
    rng = np.random.default_rng(12345)
    # (an x-slab of a decomposed run takes its rows of the whole grid's random field)
    pert = (2.0 * rng.random(size=(g.nx_global + 2 * g.ng, g.qy, 2)) - 1)[g.i0:g.i0 + g.qx]
    with np.errstate(invalid="ignore", divide="ignore"):
        cs = np.sqrt(gamma * p2d / np.asarray(dens))
    pert[:, :, 0] *= 0.05 * cs
    pert[:, :, 1] *= 0.05 * cs
    idx = np.asarray(dens) > 2 * cutoff
    xmom[idx] = np.asarray(dens)[idx] * pert[idx, 0]
    ymom[idx] = np.asarray(dens)[idx] * pert[idx, 1]
    ener = my_data.get_var("energy")
    ener[:, :] += 0.5 * (xmom[:, :]**2 + ymom[:, :]**2) / dens[:, :]


def heating_profile(myg, rp):
    dist = np.abs(np.asarray(myg.y2d) - rp.get_param("convection.y_height"))
    return rp.get_param("convection.e_rate"), \
        np.exp(-(dist / rp.get_param("convection.thickness"))**2)


def source_terms(myg, U, ivars, rp):
    rate, prof = heating_profile(myg, rp)
    S = myg.scratch_array(nvar=ivars.nvar)
    S[:, :, ivars.iener] = U[:, :, ivars.idens] * rate * prof
    return S


def finalize():
    pass
