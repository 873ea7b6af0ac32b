"""User boundaries of the compressible solver: "hse" and "ambient"
(API of pyro/compressible/BC.py:21-176; the double-Mach "ramp" boundary is not
carried).

For the conserved state both types are filled by the device kernel
k_fill_y_user (csrc/ctx.hip) through pyrohip_fill_bc; `user` below is the host
fallback with the same semantics for any other CellCenterData2d that uses
these boundary names (e.g. source-term arrays), operating on the host copy.
"""
import numpy as np

from ..util import msg
from . import eos

_COPIED = ("density", "x-momentum", "y-momentum", "dens_src", "xmom_src",
           "ymom_src", "E_src", "fuel", "ash")


def _ghost_rows(myg, bc_edge):
    """(interior row next to the edge, ghost rows walking away from it)"""
    if bc_edge == "ylb":
        return myg.jlo, range(myg.jlo - 1, -1, -1), -1.0
    return myg.jhi, range(myg.jhi + 1, myg.jhi + myg.ng + 1), 1.0


def user(bc_name, bc_edge, variable, ccdata):
    """fill the ghost cells of `variable` on edge `bc_edge` ("ylb"/"yrb")"""
    myg = ccdata.grid
    if bc_name == "hse":
        if bc_edge not in ("ylb", "yrb"):
            msg.fail("error: hse BC not supported for xlb or xrb")
        jb, rows, sgn = _ghost_rows(myg, bc_edge)
        v = ccdata.get_var(variable)
        if variable in _COPIED:
            for j in rows:
                v[:, j] = v[:, jb]
        elif variable == "energy":
            # constant density and kinetic energy, dp = rho g dy per ghost row
            dens = ccdata.get_var("density")[:, jb]
            mx = ccdata.get_var("x-momentum")[:, jb]
            my = ccdata.get_var("y-momentum")[:, jb]
            grav, gamma = ccdata.get_aux("grav"), ccdata.get_aux("gamma")
            ke = 0.5 * (mx**2 + my**2) / dens
            pres = np.array(eos.pres(gamma, dens, (v[:, jb] - ke) / dens))
            for j in rows:
                pres = pres + sgn * grav * dens * myg.dy
                v[:, j] = eos.rhoe(gamma, pres) + ke
        else:
            raise NotImplementedError("variable not defined")
    elif bc_name == "ambient":
        if bc_edge != "yrb":
            msg.fail("error: ambient BC not supported for xlb, xrb, or ylb")
        rho, u, vel, p = (ccdata.get_aux(k) for k in
                          ("ambient_rho", "ambient_u", "ambient_v", "ambient_p"))
        v = ccdata.get_var(variable)
        top = slice(myg.jhi + 1, myg.jhi + myg.ng + 1)
        v[:, top] = v[:, myg.jhi][:, np.newaxis]
        const = {"density": rho, "x-momentum": rho * u, "y-momentum": rho * vel,
                 "energy": p / (ccdata.get_aux("gamma") - 1.0) + 0.5 * rho * (u**2 + vel**2)}
        if variable in const:
            v[:, top] = const[variable]
    else:
        msg.fail(f"error: bc type {bc_name} not supported")
