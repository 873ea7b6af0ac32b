"""Compressible Euler, unsplit CTU + HLLC; `Simulation`, `Variables`,
`cons_to_prim`, `prim_to_cons` with the surface of pyro.compressible.  The
time step runs in csrc/compressible.hip / csrc/comp_fused.hip."""
from .simulation import (Simulation, Variables, cons_to_prim, prim_to_cons)

__all__ = ["Simulation", "Variables", "cons_to_prim", "prim_to_cons"]
