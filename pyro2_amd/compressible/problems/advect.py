"""A Gaussian density bump in pressure equilibrium, advected diagonally at
u = v = 1 on a Cartesian grid or in the theta direction on a SphericalPolar
grid (reference: pyro/compressible/problems/advect.py)."""
import numpy as np

from ...util import msg

DEFAULT_INPUTS = "inputs.advect.64"
PROBLEM_PARAMS = {}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the advect problem...")
    gamma = rp.get_param("eos.gamma")
    xc = 0.5 * (rp.get_param("mesh.xmin") + rp.get_param("mesh.xmax"))
    yc = 0.5 * (rp.get_param("mesh.ymin") + rp.get_param("mesh.ymax"))
    g = my_data.grid
    dens = my_data.get_var("density")
    xmom, ymom = my_data.get_var("x-momentum"), my_data.get_var("y-momentum")
    if g.coord_type == 0:
        dens[:, :] = 1.0 + np.exp(-60.0 * ((np.asarray(g.x2d) - xc)**2 +
                                           (np.asarray(g.y2d) - yc)**2))
        u = v = 1.0
    else:   # advect.py:58-74: the bump sits at mid radius, a quarter of the theta sum
        ysum = rp.get_param("mesh.ymin") + rp.get_param("mesh.ymax")
        xctr, yctr = xc * np.sin(ysum * 0.25), xc * np.cos(ysum * 0.25)
        x = np.asarray(g.x2d) * np.sin(np.asarray(g.y2d))
        y = np.asarray(g.x2d) * np.cos(np.asarray(g.y2d))
        dens[:, :] = 1.0 + np.exp(-120.0 * ((x - xctr)**2 + (y - yctr)**2))
        u, v = 0.0, 1.0
    xmom[:, :] = dens[:, :] * u
    ymom[:, :] = dens[:, :] * v
    pres = 1.0
    my_data.get_var("energy")[:, :] = pres / (gamma - 1.0) + 0.5 * (xmom**2 + ymom**2) / dens


def finalize():
    pass
