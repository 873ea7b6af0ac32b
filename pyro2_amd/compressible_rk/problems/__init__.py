"""The problem setups are those of the compressible solver (the reference
keeps copies of the same files under compressible_rk/problems)."""
import importlib
import pkgutil
import sys

from ...compressible import problems as _base

for _m in pkgutil.iter_modules(_base.__path__):
    sys.modules[f"{__name__}.{_m.name}"] = importlib.import_module(f"{_base.__name__}.{_m.name}")
