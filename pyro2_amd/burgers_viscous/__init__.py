"""Viscous Burgers solver; `Simulation` has the surface of
pyro.burgers_viscous.Simulation.  The unsplit Burgers predictor (with the
diffusion correction of the edge states) and the advective terms run in
csrc/incompressible.hip; each velocity component is then advanced by a
Crank-Nicolson Helmholtz solve in the multigrid V-cycle of csrc/multigrid.hip."""
from .simulation import Simulation

__all__ = ["Simulation"]
