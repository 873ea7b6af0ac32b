"""Uniform gas at rest that is slowly heated at the centre of the domain: a
test of the energy source term (reference: pyro/compressible/problems/heating.py).
The source S[energy] = rho e_rate exp(-(r / r_src)^2) runs on the device
through `heating_profile`; `source_terms` is the reference's host-side form."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.heating"
PROBLEM_PARAMS = {"heating.rho_ambient": 1.0, "heating.p_ambient": 10.0,
                  "heating.r_src": 0.1, "heating.e_rate": 0.1}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the heating problem...")
    gamma = rp.get_param("eos.gamma")
    my_data.get_var("density")[:, :] = rp.get_param("heating.rho_ambient")
    my_data.get_var("x-momentum")[:, :] = 0.0
    my_data.get_var("y-momentum")[:, :] = 0.0
    my_data.get_var("energy")[:, :] = rp.get_param("heating.p_ambient") / (gamma - 1.0)


def heating_profile(myg, rp):
    """(e_rate, exp(-(r/r_src)^2)) on the whole grid, r measured from the centre"""
    xc, yc = 0.5 * (myg.xmin + myg.xmax), 0.5 * (myg.ymin + myg.ymax)
    dist = np.sqrt((np.asarray(myg.x2d) - xc)**2 + (np.asarray(myg.y2d) - yc)**2)
    return rp.get_param("heating.e_rate"), np.exp(-(dist / rp.get_param("heating.r_src"))**2)


def source_terms(myg, U, ivars, rp):
    rate, prof = heating_profile(myg, rp)
    S = myg.scratch_array(nvar=ivars.nvar)
    S[:, :, ivars.iener] = U[:, :, ivars.idens] * rate * prof
    return S


def finalize():
    pass
