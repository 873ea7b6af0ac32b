"""A shock moving along the diagonal: (u, v) = (3, 3) below the line
x + y = 1 and (1, 1) above it (reference: pyro/burgers/problems/test.py)."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.test"
PROBLEM_PARAMS = {}


def init_data(myd, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the burgers test problem...")
    g = myd.grid
    above = np.asarray(g.y2d) > -1.0 * np.asarray(g.x2d) + 1.0
    for name in ("x-velocity", "y-velocity"):
        myd.get_var(name)[:, :] = np.where(above, 1.0, 3.0)


def finalize():
    pass
