r"""Smooth viscous convergence test (pyro/incompressible_viscous/problems/
converge.py:1-70): the incompressible converge field, whose exact solution
decays as exp(-8 pi^2 nu t) while it is advected with (1, 1)."""
import math

import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.converge.64"

PROBLEM_PARAMS = {}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the incompressible viscous converge problem...")
    u = my_data.get_var("x-velocity")
    v = my_data.get_var("y-velocity")
    myg = my_data.grid
    if myg.xmin != 0 or myg.xmax != 1 or myg.ymin != 0 or myg.ymax != 1:
        msg.fail("ERROR: domain should be a unit square")
    u[:, :] = 1.0 - 2.0 * np.cos(2.0 * math.pi * myg.x2d) * np.sin(2.0 * math.pi * myg.y2d)
    v[:, :] = 1.0 + 2.0 * np.sin(2.0 * math.pi * myg.x2d) * np.cos(2.0 * math.pi * myg.y2d)


def exact(myg, t, nu):
    """u, v of the exact solution at time t (module docstring of the reference)"""
    decay = math.exp(-8.0 * math.pi**2 * nu * t)
    u = 1.0 - 2.0 * np.cos(2.0 * math.pi * (myg.x2d - t)) * \
        np.sin(2.0 * math.pi * (myg.y2d - t)) * decay
    v = 1.0 + 2.0 * np.sin(2.0 * math.pi * (myg.x2d - t)) * \
        np.cos(2.0 * math.pi * (myg.y2d - t)) * decay
    return u, v


def finalize():
    print("""
          Comparisons to the analytic solution: problems.converge.exact(grid, t, nu)
          """)
