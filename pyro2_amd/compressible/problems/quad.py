"""Four-quadrant 2-d Riemann problem, configuration 3 of Schulz-Rinne et al.
1993 / Lax & Liu 1998 (reference: pyro/compressible/problems/quad.py)."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.quad"
PROBLEM_PARAMS = {
    "quadrant.rho1": 1.5, "quadrant.u1": 0.0, "quadrant.v1": 0.0, "quadrant.p1": 1.5,
    "quadrant.rho2": 0.532258064516129, "quadrant.u2": 1.206045378311055,
    "quadrant.v2": 0.0, "quadrant.p2": 0.3,
    "quadrant.rho3": 0.137992831541219, "quadrant.u3": 1.206045378311055,
    "quadrant.v3": 1.206045378311055, "quadrant.p3": 0.029032258064516,
    "quadrant.rho4": 0.532258064516129, "quadrant.u4": 0.0,
    "quadrant.v4": 1.206045378311055, "quadrant.p4": 0.3,
    "quadrant.cx": 0.5, "quadrant.cy": 0.5}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the quadrant problem...")
    gamma = rp.get_param("eos.gamma")
    cx, cy = rp.get_param("quadrant.cx"), rp.get_param("quadrant.cy")
    g = my_data.grid
    right, top = g.x2d >= cx, g.y2d >= cy
    masks = {1: np.logical_and(right, top), 2: np.logical_and(~right, top),
             3: np.logical_and(~right, ~top), 4: np.logical_and(right, ~top)}
    dens = my_data.get_var("density")
    xmom = my_data.get_var("x-momentum")
    ymom = my_data.get_var("y-momentum")
    ener = my_data.get_var("energy")
    for q, m in masks.items():
        r, u, v, p = (rp.get_param(f"quadrant.{k}{q}") for k in ("rho", "u", "v", "p"))
        dens[m] = r
        xmom[m] = r * u
        ymom[m] = r * v
        ener[m] = p / (gamma - 1.0) + 0.5 * r * (u * u + v * v)


def finalize():
    pass
