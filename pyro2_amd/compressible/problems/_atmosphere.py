"""Adiabatically stratified atmosphere shared by the plume and convection
problems: rho(y) = rho_base (1 - (gamma-1)/gamma y/H)^(1/(gamma-1)) with a
density floor; helper, not a problem module."""
import numpy as np


def adiabatic_density(g, gamma, dens_base, scale_height, dens_cutoff):
    """density of the interior rows (1-d over j; ghost rows keep the cut-off)"""
    rho = np.full(g.qy, dens_cutoff, dtype=np.float64)
    for j in range(g.jlo, g.jhi + 1):
        prof = 1.0 - (gamma - 1.0) / gamma * g.y[j] / scale_height
        if prof > 0.0:
            rho[j] = max(dens_base * prof**(1.0 / (gamma - 1.0)), dens_cutoff)
    return rho
