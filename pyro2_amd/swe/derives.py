"""derived fields of the shallow-water state (pyro/swe/derives.py:4-42)"""
import numpy as np


def derive_primitives(myd, varnames):
    h = myd.get_var("height")
    u = myd.get_var("x-momentum") / h
    v = myd.get_var("y-momentum") / h
    g = myd.get_aux("g")
    wanted = [varnames] if isinstance(varnames, str) else list(varnames)
    table = {"velocity": (u, v), "primitive": (h, u, v)}
    out = []
    for var in wanted:
        if var == "soundspeed":
            out.append(np.sqrt(g * h))
        else:
            out.extend(table.get(var, ()))
    return out if len(out) > 1 else out[0]
