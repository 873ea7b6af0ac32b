"""Incompressible flow with constant kinematic viscosity; `Simulation` has the
surface of pyro.incompressible_viscous.Simulation.  On top of the
incompressible solver: the viscous source in the CTU predictor and two
Crank-Nicolson Helmholtz solves per step (one per velocity component) in the
multigrid V-cycle of csrc/multigrid.hip, i.e. four MG solves per step; the
"moving_lid" boundary of the lid-driven cavity is a device ghost-fill type."""
from .simulation import Simulation

__all__ = ["Simulation"]
