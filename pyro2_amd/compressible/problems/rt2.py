"""Rayleigh-Taylor instability seeded with two different wavelengths: 18 waves
over the left third of the domain, 3 over the rest (reference:
pyro/compressible/problems/rt2.py)."""
import numpy as np

from ...util import msg
from ._stratified import finish, two_layer

DEFAULT_INPUTS = "inputs.rt2"
PROBLEM_PARAMS = {"rt2.dens1": 1.0, "rt2.dens2": 2.0, "rt2.amp": 1.0, "rt2.sigma": 0.1,
                  "rt2.p0": 10.0}
_WAVES_LEFT, _WAVES_RIGHT = 18, 3


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the rt problem...")
    g = my_data.grid
    amp, sigma = rp.get_param("rt2.amp"), rp.get_param("rt2.sigma")
    rho, p, ymid = two_layer(g, rp.get_param("rt2.dens1"), rp.get_param("rt2.dens2"),
                             rp.get_param("rt2.p0"), rp.get_param("compressible.grav"))
    x2d, y2d = np.asarray(g.x2d), np.asarray(g.y2d)
    L = g.xmax - g.xmin
    left = x2d < L / 3.0
    envelope = np.exp(-(y2d - ymid)**2 / sigma**2)
    vel = np.where(left, amp * np.sin(4.0 * np.pi * _WAVES_LEFT * x2d / L) * envelope,
                   amp * np.sin(4.0 * np.pi * _WAVES_RIGHT * x2d / L) * envelope)
    finish(my_data, rho, p, vel, rp.get_param("eos.gamma"))


def finalize():
    pass
