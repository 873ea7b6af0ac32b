"""Rayleigh-Taylor instability seeded with nmodes cosine modes of random phase
and amplitude (fixed seed) (reference: pyro/compressible/problems/
rt_multimode.py)."""
import numpy as np

from ...util import msg
from ._stratified import finish, two_layer

DEFAULT_INPUTS = "inputs.rt_multimode"
PROBLEM_PARAMS = {"rt_multimode.dens1": 1.0, "rt_multimode.dens2": 2.0, "rt_multimode.amp": 1.0,
                  "rt_multimode.sigma": 0.1, "rt_multimode.nmodes": 10, "rt_multimode.p0": 10.0}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the rt problem...")
    get = lambda k: rp.get_param("rt_multimode." + k)   # noqa: E731
    g = my_data.grid
    rho, p, ymid = two_layer(g, get("dens1"), get("dens2"), get("p0"),
                             rp.get_param("compressible.grav"))
    x2d, y2d = np.asarray(g.x2d), np.asarray(g.y2d)
    L = g.xmax - g.xmin
    amp, sigma, nmodes = get("amp"), get("sigma"), get("nmodes")
    rng = np.random.default_rng(12345)
    vel = np.zeros_like(x2d)
    for k in range(1, nmodes + 1):
        phase = rng.random() * 2 * np.pi
        mode_amp = amp * rng.random()
        vel += mode_amp * np.cos(2.0 * np.pi * k * x2d / L + phase) * \
            np.exp(-(y2d - ymid)**2 / sigma**2)
    vel /= nmodes
    finish(my_data, rho, p, vel, rp.get_param("eos.gamma"))


def finalize():
    pass
