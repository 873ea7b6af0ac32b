"""Implicit (Crank-Nicolson) diffusion: a caller of the device multigrid
solver through the MG.CellCenterMG2d machinery, `Simulation` with the surface
of pyro.diffusion.Simulation."""
from .simulation import Simulation

__all__ = ["Simulation"]
