"""Acoustic pulse of McCorquodale & Colella (2011): a smooth, compactly
supported density bump at rest in an isentropic gas (reference:
pyro/compressible/problems/acoustic_pulse.py)."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.acoustic_pulse"
PROBLEM_PARAMS = {"acoustic_pulse.rho0": 1.4, "acoustic_pulse.drho0": 0.14}


def init_data(myd, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the acoustic pulse problem...")
    gamma = rp.get_param("eos.gamma")
    rho0, drho0 = rp.get_param("acoustic_pulse.rho0"), rp.get_param("acoustic_pulse.drho0")
    xc = 0.5 * (rp.get_param("mesh.xmin") + rp.get_param("mesh.xmax"))
    yc = 0.5 * (rp.get_param("mesh.ymin") + rp.get_param("mesh.ymax"))
    g = myd.grid
    dist = np.sqrt((np.asarray(g.x2d) - xc)**2 + (np.asarray(g.y2d) - yc)**2)
    dens = myd.get_var("density")
    bump = rho0 + drho0 * np.exp(-16 * dist**2) * np.cos(np.pi * dist)**6
    dens[:, :] = np.where(dist <= 0.5, bump, rho0)
    myd.get_var("x-momentum")[:, :] = 0.0
    myd.get_var("y-momentum")[:, :] = 0.0
    myd.get_var("energy")[:, :] = (np.asarray(dens) / rho0)**gamma / (gamma - 1)


def finalize():
    pass
