"""Import alias: code written against python-hydro/pyro2 (`from pyro.pyro_sim
import Pyro`, `import pyro.multigrid.MG as MG`, `from pyro.mesh import patch`,
...) runs unchanged on the MI355X path.  Every `pyro.X` resolves to the module
object `pyro2_amd.X` (the drop-in boundary is pyro's module and
class surface).  Nothing is implemented here."""
import importlib
import importlib.abc
import importlib.util
import sys

import pyro2_amd

_SRC, _DST = "pyro2_amd", __name__


class _AliasLoader(importlib.abc.Loader):
    def create_module(self, spec):
        return importlib.import_module(_SRC + spec.name[len(_DST):])

    def exec_module(self, module):      # already executed under its real name
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_DST + "."):
            return None
        real = _SRC + fullname[len(_DST):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader())


sys.meta_path.insert(0, _AliasFinder())
# the top-level names of the package (Pyro, ...) are reachable as pyro.<name>
globals().update({k: v for k, v in vars(pyro2_amd).items() if not k.startswith("__")})


def __getattr__(name):       # lazily resolved exports of the package (Pyro, CellCenterData2d, ...)
    return getattr(pyro2_amd, name)
