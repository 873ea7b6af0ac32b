"""Method-of-lines compressible solver (piecewise-linear reconstruction,
one Riemann problem per face, Runge-Kutta in time); `Simulation` has the
surface of pyro.compressible_rk.Simulation.  The right-hand side runs in
csrc/compressible.hip (k_rk_*), the stage algebra in pyrohip_state_lincomb."""
from .simulation import Simulation

__all__ = ["Simulation"]
