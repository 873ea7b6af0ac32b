"""Derived variables for analysis / plotting (velocity, pressure, soundspeed,
...), the get_var() callback of pyro/compressible/derives.py:6-69.  Host-side
NumPy on a downloaded copy: this is post-processing, not the hot path (the
hot path's CFL reduction runs on the device, pyrohip_comp_dt)."""
import numpy as np

from . import eos


def derive_primitives(myd, varnames):
    dens = myd.get_var("density")
    xmom = myd.get_var("x-momentum")
    ymom = myd.get_var("y-momentum")
    ener = myd.get_var("energy")
    u = xmom / dens
    v = ymom / dens
    e = (ener - 0.5 * dens * (u * u + v * v)) / dens
    gamma = myd.get_aux("gamma")
    p = eos.pres(gamma, dens, e)
    wanted = [varnames] if isinstance(varnames, str) else list(varnames)
    out = []
    for var in wanted:
        if var == "velocity":
            out += [u, v]
        elif var in ("e", "eint"):
            out.append(e)
        elif var in ("p", "pressure"):
            out.append(p)
        elif var == "primitive":
            out += [dens, u, v, p]
        elif var == "soundspeed":
            out.append(np.sqrt(gamma * p / dens))
        elif var == "machnumber":
            out.append(np.sqrt(u**2 + v**2) / np.sqrt(gamma * p / dens))
        elif var == "vorticity":
            g = myd.grid
            vort = g.scratch_array()
            vort.v()[:, :] = 0.5 * (v.ip(1) - v.ip(-1)) / g.dx - 0.5 * (u.jp(1) - u.jp(-1)) / g.dy
            out.append(vort)
    if len(out) > 1:
        return out
    return out[0]
