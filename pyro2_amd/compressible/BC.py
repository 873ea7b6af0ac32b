"""User boundaries of the compressible solver: "hse", "ambient" and the "ramp"
boundary of the double Mach reflection problem (API of
pyro/compressible/BC.py:21-296).

For the conserved state all three are filled by device kernels (k_fill_y_user,
k_fill_ramp in csrc/ctx.hip) through pyrohip_fill_bc; `user` below is the host
fallback with the same semantics for any other CellCenterData2d that uses
these boundary names (e.g. source-term arrays), operating on the host copy.
"""
import math

import numpy as np

from ..util import msg
from . import eos

_COPIED = ("density", "x-momentum", "y-momentum", "dens_src", "xmom_src",
           "ymom_src", "E_src", "fuel", "ash")


# post- and pre-shock states of the Mach 10 shock (BC.py:254-296)
_POST = dict(r=8.0, u=7.1447096, v=-4.125, p=116.5)
_PRE = dict(r=1.4, u=0.0, v=0.0, p=1.0)


def _inflow(state, gamma):
    """conserved values in the state's order density, energy, x-, y-momentum"""
    r, u, v, p = state["r"], state["u"], state["v"], state["p"]
    return np.array([r, p / (gamma - 1.0) + 0.5 * r * (u * u + v * v), r * u, r * v])


def ramp_params(myg, gamma, t):
    """what the device fill needs for the "ramp" boundary at time t: cell
    centres, the sub-sampling offset 0.5 dx sqrt(3), the inflow states and the
    positions of the shock front on the ghost rows above the upper y boundary
    (a Mach 10 shock at 60 degrees to the x axis that started at x = 1/6,
    BC.py:240-243).  Evaluated here with math.* exactly like the reference."""
    slope, speed = math.tan(math.pi / 3.0), 10.0 / math.sin(math.pi / 3.0)
    sfd, sfu = np.zeros(8), np.zeros(8)
    for k in range(myg.ng):
        yj = float(myg.y[myg.jhi + 1 + k])
        sfu[k] = 1.0 / 6.0 + (yj + 0.5 * myg.dy * math.sqrt(3)) / slope + speed * t
        sfd[k] = 1.0 / 6.0 + (yj - 0.5 * myg.dy * math.sqrt(3)) / slope + speed * t
    return dict(x=np.ascontiguousarray(myg.x, dtype=np.float64),
                cxoff=0.5 * myg.dx * math.sqrt(3), post=_inflow(_POST, gamma),
                pre=_inflow(_PRE, gamma), sf_down=sfd, sf_up=sfu)


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
    elif bc_name == "ramp":
        names = ("density", "energy", "x-momentum", "y-momentum")
        v = ccdata.get_var(variable)
        if variable not in names:
            v[:, :] = 0.0            # no source term
            return
        n = names.index(variable)
        rp = ramp_params(myg, ccdata.get_aux("gamma"), ccdata.t)
        post, pre = rp["post"][n], rp["pre"][n]
        if bc_edge == "xlb":
            v[:myg.ilo, :] = post
        elif bc_edge == "ylb":
            left = np.asarray(myg.x) < 1.0 / 6.0
            for jj, j in enumerate(range(myg.jlo - 1, -1, -1)):
                mirror = v[:, myg.jlo + jj]
                v[:, j] = np.where(left, post, -1.0 * mirror if variable == "y-momentum" else mirror)
        elif bc_edge == "yrb":
            x = np.asarray(myg.x)
            for k in range(myg.ng):
                acc = np.zeros_like(x)
                for sf in (rp["sf_down"][k], rp["sf_up"][k]):
                    for cx in (x - rp["cxoff"], x + rp["cxoff"]):
                        acc = acc + 0.25 * np.where(cx < sf, post, pre)
                v[:, myg.jhi + 1 + k] = acc
    else:
        msg.fail(f"error: bc type {bc_name} not supported")
