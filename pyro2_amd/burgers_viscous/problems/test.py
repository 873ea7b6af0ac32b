"""The burgers `test` set-up (pyro/burgers_viscous/problems/test.py is that file)."""
from ...burgers.problems.test import PROBLEM_PARAMS, finalize, init_data  # noqa: F401

DEFAULT_INPUTS = "inputs.test"
