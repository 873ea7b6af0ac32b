"""Doubly periodic shear layer with viscosity: the incompressible problem of the
same name (pyro/incompressible_viscous/problems/shear.py is that setup)."""
from ...incompressible.problems.shear import PROBLEM_PARAMS, finalize, init_data  # noqa: F401

DEFAULT_INPUTS = "inputs.shear"
