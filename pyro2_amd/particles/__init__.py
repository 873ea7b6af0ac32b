"""Passive tracer particles (pyro/particles)."""
from . import particles  # noqa: F401
