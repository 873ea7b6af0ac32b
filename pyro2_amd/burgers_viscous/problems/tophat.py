"""A circular "top hat" of unit velocity in a fluid at rest: it steepens into
a shock that the viscosity smears (pyro/burgers_viscous/problems/tophat.py)."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.tophat"
PROBLEM_PARAMS = {}


def init_data(myd, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the tophat burgers problem...")
    g = myd.grid
    xctr, yctr = 0.5 * (g.xmin + g.xmax), 0.5 * (g.ymin + g.ymax)
    R = 0.1
    inside = (np.asarray(g.x2d) - xctr)**2 + (np.asarray(g.y2d) - yctr)**2 < R**2
    for name in ("x-velocity", "y-velocity"):
        a = myd.get_var(name)
        a[:, :] = 0.0
        a[inside] = 1.0


def finalize():
    pass
