"""Disc of radius 0.1 with value 1 on a zero background (reference:
pyro/advection/problems/tophat.py); hammers the limiters."""
from ...util import msg

DEFAULT_INPUTS = "inputs.tophat"
PROBLEM_PARAMS = {}


def init_data(myd, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the tophat advection problem...")
    g = myd.grid
    xc = 0.5 * (g.xmin + g.xmax)
    yc = 0.5 * (g.ymin + g.ymax)
    dens = myd.get_var("density")
    dens[:, :] = 0.0
    R = 0.1
    dens[(g.x2d - xc)**2 + (g.y2d - yc)**2 < R**2] = 1.0


def finalize():
    pass
