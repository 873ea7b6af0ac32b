"""Dam break along x or y: still water of depth h_left next to depth h_right
(reference: pyro/swe/problems/dam.py)."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.dam.x"
PROBLEM_PARAMS = {"dam.direction": "x", "dam.h_left": 1.0, "dam.h_right": 0.125,
                  "dam.u_left": 0.0, "dam.u_right": 0.0}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the dam problem...")
    left = (rp.get_param("dam.h_left"), rp.get_param("dam.u_left"), 1.0)
    right = (rp.get_param("dam.h_right"), rp.get_param("dam.u_right"), 0.0)
    g = my_data.grid
    if rp.get_param("dam.direction") == "x":
        coord = np.asarray(g.x2d)
        ctr = 0.5 * (rp.get_param("mesh.xmin") + rp.get_param("mesh.xmax"))
        mom_n, mom_t = "x-momentum", "y-momentum"
    else:
        coord = np.asarray(g.y2d)
        ctr = 0.5 * (rp.get_param("mesh.ymin") + rp.get_param("mesh.ymax"))
        mom_n, mom_t = "y-momentum", "x-momentum"
    h, X = my_data.get_var("height"), my_data.get_var("fuel")
    mn, mt = my_data.get_var(mom_n), my_data.get_var(mom_t)
    for mask, (hh, uu, xx) in ((coord <= ctr, left), (coord > ctr, right)):
        h[mask] = hh
        mn[mask] = hh * uu
        mt[mask] = 0.0
        X[mask] = xx


def finalize():
    pass
