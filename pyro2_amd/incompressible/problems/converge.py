"""Smooth travelling-wave solution of the incompressible Euler equations
(Minion 1996) for convergence tests:
    u = 1 - 2 cos(2 pi (x - t)) sin(2 pi (y - t)),
    v = 1 + 2 sin(2 pi (x - t)) cos(2 pi (y - t)),
    p = -cos(4 pi (x - t)) - cos(4 pi (y - t)).
Reference: pyro/incompressible/problems/converge.py."""
import math

import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.converge.64"
PROBLEM_PARAMS = {}


def exact(g, t):
    """(u, v) of the analytic solution at time t on the full arrays of grid g"""
    x, y = 2.0 * math.pi * (np.asarray(g.x2d) - t), 2.0 * math.pi * (np.asarray(g.y2d) - t)
    return 1.0 - 2.0 * np.cos(x) * np.sin(y), 1.0 + 2.0 * np.sin(x) * np.cos(y)


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the incompressible converge problem...")
    g = my_data.grid
    if (g.xmin, g.xmax, g.ymin, g.ymax) != (0, 1, 0, 1):
        msg.fail("ERROR: domain should be a unit square")
    x2d, y2d = np.asarray(g.x2d), np.asarray(g.y2d)
    my_data.get_var("x-velocity")[:, :] = \
        1.0 - 2.0 * np.cos(2.0 * math.pi * x2d) * np.sin(2.0 * math.pi * y2d)
    my_data.get_var("y-velocity")[:, :] = \
        1.0 + 2.0 * np.sin(2.0 * math.pi * x2d) * np.cos(2.0 * math.pi * y2d)


def finalize():
    print("""
          The analytic solution is available as
          pyro2_amd.incompressible.problems.converge.exact(grid, t).
          """)
