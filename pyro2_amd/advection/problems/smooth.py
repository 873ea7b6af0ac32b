"""Gaussian bump on a unit background (the reference's advection `smooth`
problem, pyro/advection/problems/smooth.py:14-32): smooth, so the limiters
stay quiet -- the convergence-test problem."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.smooth"
PROBLEM_PARAMS = {}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the smooth advection problem...")
    g = my_data.grid
    xc = 0.5 * (g.xmin + g.xmax)
    yc = 0.5 * (g.ymin + g.ymax)
    dens = my_data.get_var("density")
    dens[:, :] = 1.0 + np.exp(-60.0 * ((g.x2d - xc)**2 + (g.y2d - yc)**2))


def finalize():
    pass
