"""Smooth velocity field for convergence tests: both components are a constant
plus a Gaussian (pyro/burgers_viscous/problems/converge.py)."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.converge.64"
PROBLEM_PARAMS = {}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the smooth burgers convergence problem...")
    g = my_data.grid
    xctr, yctr = 0.5 * (g.xmin + g.xmax), 0.5 * (g.ymin + g.ymax)
    A = 0.05   # magnitude of the field
    bump = A + A * np.exp(-50.0 * ((np.asarray(g.x2d) - xctr)**2 + (np.asarray(g.y2d) - yctr)**2))
    my_data.get_var("x-velocity")[:, :] = bump
    my_data.get_var("y-velocity")[:, :] = bump


def finalize():
    pass
