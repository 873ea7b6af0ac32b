"""Four-quadrant Riemann problem for the shallow-water equations: constant
states (h, u, v) in the quadrants around the corner (cx, cy); the fuel tracer
marks quadrants 1 and 3 (reference: pyro/swe/problems/quad.py)."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.quad"
PROBLEM_PARAMS = {"quadrant.h1": 1.5, "quadrant.u1": 0.0, "quadrant.v1": 0.0,
                  "quadrant.h2": 0.532258064516129, "quadrant.u2": 1.206045378311055,
                  "quadrant.v2": 0.0,
                  "quadrant.h3": 0.137992831541219, "quadrant.u3": 1.206045378311055,
                  "quadrant.v3": 1.206045378311055,
                  "quadrant.h4": 0.532258064516129, "quadrant.u4": 0.0,
                  "quadrant.v4": 1.206045378311055,
                  "quadrant.cx": 0.5, "quadrant.cy": 0.5}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the quadrant problem...")
    cx, cy = rp.get_param("quadrant.cx"), rp.get_param("quadrant.cy")
    g = my_data.grid
    right, top = np.asarray(g.x2d) >= cx, np.asarray(g.y2d) >= cy
    masks = {1: right & top, 2: ~right & top, 3: ~right & ~top, 4: right & ~top}
    h, X = my_data.get_var("height"), my_data.get_var("fuel")
    xmom, ymom = my_data.get_var("x-momentum"), my_data.get_var("y-momentum")
    for k, mask in masks.items():
        hk, uk, vk = (rp.get_param(f"quadrant.{c}{k}") for c in "huv")
        h[mask] = hk
        xmom[mask] = hk * uk
        ymom[mask] = hk * vk
        X[mask] = 1.0 if k in (1, 3) else 0.0
    X *= h


def finalize():
    pass
