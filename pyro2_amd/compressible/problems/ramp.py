"""Double Mach reflection (Woodward & Colella 1984): a Mach 10 shock at 60
degrees to a reflecting wall that starts at x = 1/6.  Cells cut by the initial
shock front get the average of four sub-samples.  The matching time-dependent
boundary is the "ramp" type of compressible/BC.py.  Reference:
pyro/compressible/problems/ramp.py."""
import math

import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.ramp"
PROBLEM_PARAMS = {"ramp.rhol": 8.0, "ramp.ul": 7.1447096, "ramp.vl": -4.125, "ramp.pl": 116.5,
                  "ramp.rhor": 1.4, "ramp.ur": 0.0, "ramp.vr": 0.0, "ramp.pr": 1.0}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the double Mach reflection problem...")
    gamma = rp.get_param("eos.gamma")
    states = {}
    for side in "lr":
        r, u, v, p = (rp.get_param(f"ramp.{k}{side}") for k in ("rho", "u", "v", "p"))
        states[side] = (r, r * u, r * v, p / (gamma - 1.0) + 0.5 * r * (u * u + v * v))
    g = my_data.grid
    names = ("density", "x-momentum", "y-momentum", "energy")
    fields = [my_data.get_var(n) for n in names]
    fields[0][:, :] = 1.4
    I = (slice(g.ilo, g.ihi + 1), slice(g.jlo, g.jhi + 1))
    x = np.asarray(g.x)[I[0], np.newaxis]
    y = np.asarray(g.y)[np.newaxis, I[1]]
    off_x, off_y = 0.5 * g.dx * math.sqrt(3), 0.5 * g.dy * math.sqrt(3)
    slope = math.tan(math.pi / 3.0)
    fronts = (slope * (x - off_x - 1.0 / 6.0), slope * (x + off_x - 1.0 / 6.0))
    acc = [np.zeros((g.nx, g.ny)) for _ in names]
    for ys in (y - off_y, y + off_y):          # same order of accumulation as the reference
        for front in fronts:
            behind = ys >= front
            for a, vl, vr in zip(acc, states["l"], states["r"]):
                a += np.where(behind, 0.25 * vl, 0.25 * vr)
    for f, a in zip(fields, acc):
        f[I] = a


def finalize():
    pass
