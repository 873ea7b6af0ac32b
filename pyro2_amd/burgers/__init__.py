"""Unsplit CTU solver for the 2-d inviscid Burgers equation; `Simulation` has
the surface of pyro.burgers.Simulation, the update runs in
csrc/incompressible.hip (the predictor it shares with the incompressible
solver)."""
from .simulation import Simulation

__all__ = ["Simulation"]
