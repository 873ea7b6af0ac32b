"""Shock tube along x or y (reference: pyro/compressible/problems/sod.py)."""
from ...util import msg

DEFAULT_INPUTS = "inputs.sod.x"
PROBLEM_PARAMS = {"sod.direction": "x",
                  "sod.dens_left": 1.0, "sod.dens_right": 0.125,
                  "sod.u_left": 0.0, "sod.u_right": 0.0,
                  "sod.p_left": 1.0, "sod.p_right": 0.1}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the sod problem...")
    left = {k: rp.get_param("sod." + k + "_left") for k in ("dens", "u", "p")}
    right = {k: rp.get_param("sod." + k + "_right") for k in ("dens", "u", "p")}
    gamma = rp.get_param("eos.gamma")
    g = my_data.grid
    along_x = rp.get_param("sod.direction") == "x"
    if along_x:
        coord = g.x2d
        ctr = 0.5 * (rp.get_param("mesh.xmin") + rp.get_param("mesh.xmax"))
        mom_n, mom_t = "x-momentum", "y-momentum"
    else:
        coord = g.y2d
        ctr = 0.5 * (rp.get_param("mesh.ymin") + rp.get_param("mesh.ymax"))
        mom_n, mom_t = "y-momentum", "x-momentum"
    dens = my_data.get_var("density")
    ener = my_data.get_var("energy")
    mn = my_data.get_var(mom_n)
    mt = my_data.get_var(mom_t)
    for mask, s in ((coord <= ctr, left), (coord > ctr, right)):
        dens[mask] = s["dens"]
        mn[mask] = s["dens"] * s["u"]
        mt[mask] = 0.0
        ener[mask] = s["p"] / (gamma - 1.0) + 0.5 * mn[mask] * s["u"]


def finalize():
    print("""
          The exact Riemann solution for these states can be overplotted with
          pyro's analysis/sod_compare.py.
          """)
