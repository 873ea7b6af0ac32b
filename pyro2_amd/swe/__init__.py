"""Shallow-water solver (unsplit CTU, Roe or HLLC Riemann solver); the names
of pyro.swe (Simulation, Variables, cons_to_prim, prim_to_cons), the update in
csrc/swe.hip."""
from .simulation import Simulation, Variables, cons_to_prim, prim_to_cons

__all__ = ["Simulation", "Variables", "cons_to_prim", "prim_to_cons"]
