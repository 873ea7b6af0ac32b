"""Doubly periodic shear layer (Martin & Colella 2000): two tanh shear layers
at y = 1/4 and y = 3/4 in the unit square, seeded with a sinusoidal vertical
velocity.  Reference: pyro/incompressible/problems/shear.py."""
import math

import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.shear"
PROBLEM_PARAMS = {"shear.rho_s": 42.0,    # inverse width of the layers
                  "shear.delta_s": 0.05}  # amplitude of the perturbation


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the incompressible shear problem...")
    rho_s = rp.get_param("shear.rho_s")
    delta_s = rp.get_param("shear.delta_s")
    g = my_data.grid
    if (g.xmin, g.xmax, g.ymin, g.ymax) != (0, 1, 0, 1):
        msg.fail("ERROR: domain should be a unit square")
    x2d, y2d = np.asarray(g.x2d), np.asarray(g.y2d)
    y_half = 0.5 * (g.ymin + g.ymax)
    print("y_half = ", y_half)
    print("delta_s = ", delta_s)
    print("rho_s = ", rho_s)
    u = my_data.get_var("x-velocity")
    v = my_data.get_var("y-velocity")
    lower = y2d <= y_half
    u[:, :] = np.where(lower, np.tanh(rho_s * (y2d - 0.25)), np.tanh(rho_s * (0.75 - y2d)))
    v[:, :] = delta_s * np.sin(2.0 * math.pi * x2d)
    print("extrema: ", u.min(), u.max())


def finalize():
    pass
