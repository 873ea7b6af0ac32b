"""2nd-order unsplit CTU linear advection; `Simulation` has the surface of
pyro.advection.Simulation, the update runs in csrc/advection.hip."""
from .simulation import Simulation

__all__ = ["Simulation"]
