"""Approximate-projection solver for constant-density incompressible flow;
`Simulation` has the surface of pyro.incompressible.Simulation.  The CTU
predictor runs in csrc/incompressible.hip, the two elliptic solves per step in
the multigrid V-cycle of csrc/multigrid.hip; the velocity never leaves HBM
inside Pyro.run_sim."""
from .simulation import Simulation

__all__ = ["Simulation"]
