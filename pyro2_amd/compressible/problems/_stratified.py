"""Two-layer hydrostatic stratification of the Rayleigh-Taylor set-ups: fluid
of density dens1 below the mid-plane, dens2 above, pressure p0 at y = 0
integrated upwards.  Helper, not a problem module."""
import numpy as np


def two_layer(g, dens1, dens2, p0, grav):
    """1-d density and pressure over j (interior rows; 0 on the ghost rows),
    and the height of the interface"""
    ymid = 0.5 * (g.ymin + g.ymax)
    y = np.asarray(g.y)
    inside = np.zeros(g.qy, dtype=bool)
    inside[g.jlo:g.jhi + 1] = True
    lower, upper = inside & (y < ymid), inside & ~(y < ymid)
    rho = np.where(lower, dens1, np.where(upper, dens2, 0.0))
    p = np.where(lower, p0 + dens1 * grav * y,
                 np.where(upper, p0 + dens1 * grav * ymid + dens2 * grav * (y - ymid), 0.0))
    return rho, p, ymid


def finish(my_data, rho_y, p_y, ymom_velocity, gamma):
    """state from the 1-d stratification and a vertical velocity field"""
    dens = my_data.get_var("density")
    xmom, ymom = my_data.get_var("x-momentum"), my_data.get_var("y-momentum")
    dens[:, :] = rho_y[np.newaxis, :]
    xmom[:, :] = 0.0
    ymom[:, :] = ymom_velocity
    ymom *= dens
    with np.errstate(invalid="ignore", divide="ignore"):   # 0/0 in the y ghost rows
        my_data.get_var("energy")[:, :] = p_y[np.newaxis, :] / (gamma - 1.0) + \
            0.5 * (xmom**2 + ymom**2) / dens
