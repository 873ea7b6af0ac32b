"""Gaussian profile: with constant conductivity it stays Gaussian while its
peak drops and its width grows, so the run can be checked against
`phi_analytic` (reference: pyro/diffusion/problems/gaussian.py)."""
import numpy

from ...util import msg

DEFAULT_INPUTS = "inputs.gaussian"
PROBLEM_PARAMS = {"gaussian.t_0": 0.001, "gaussian.phi_0": 1.0, "gaussian.phi_max": 2.0}


def phi_analytic(dist, t, t_0, k, phi_1, phi_2):
    return (phi_2 - phi_1) * (t_0 / (t + t_0)) * \
        numpy.exp(-0.25 * dist**2 / (k * (t + t_0))) + phi_1


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the Gaussian diffusion problem...")
    g = my_data.grid
    xctr = 0.5 * (g.xmin + g.xmax)
    yctr = 0.5 * (g.ymin + g.ymax)
    k = rp.get_param("diffusion.k")
    t_0 = rp.get_param("gaussian.t_0")
    phi_max = rp.get_param("gaussian.phi_max")
    phi_0 = rp.get_param("gaussian.phi_0")
    dist = numpy.sqrt((g.x2d - xctr)**2 + (g.y2d - yctr)**2)
    my_data.get_var("phi")[:, :] = phi_analytic(dist, 0.0, t_0, k, phi_0, phi_max)
    for key, val in (("k", k), ("t_0", t_0), ("phi_0", phi_0), ("phi_max", phi_max)):
        my_data.set_aux(key, val)


def finalize():
    print("""
          The run can be compared with phi_analytic (pyro's
          analysis/gauss_diffusion_compare.py).
          """)
