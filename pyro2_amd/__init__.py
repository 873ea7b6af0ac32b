"""pyro2_amd -- the per-timestep hot path of python-hydro/pyro2 on AMD
Instinct MI355X (gfx950): hand-written HIP kernels behind pyro's own Python
class surface (Pyro / Simulation / CellCenterData2d / Grid2d, MG.CellCenterMG2d).

    from pyro2_amd import Pyro
    p = Pyro("compressible"); p.initialize_problem("sedov"); p.run_sim()

There is no CPU fallback: importing works anywhere, the first device operation
needs pyro2_amd/lib/libpyrohip.so (python -m pyro2_amd.build) and a GPU.
"""
import os as _os

__version__ = "0.1.0"

# The halo exchange of a decomposed run and the boundary strips of its steps run on a second HIP
# stream BESIDE the interior strips.  HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4) and two streams that land on one queue run one after the other -- measured on
# a 2048 x 16384 slab: 1.58 ms per step with 4 queues, 1.33 with 8 (profiles/r06_slab_queues.txt).
# Read by the HIP runtime when it initialises: set before the first device call; the user's wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


# the names pyro/__init__.py exports, resolved on first use (importing the package
# stays cheap and needs no GPU)
_LAZY = {"Pyro": ("pyro_sim", "Pyro"),
         "BC": ("mesh.boundary", "BC"), "ArrayIndexer": ("mesh.array_indexer", "ArrayIndexer"),
         "CellCenterData2d": ("mesh.patch", "CellCenterData2d"), "Grid2d": ("mesh.patch", "Grid2d"),
         "RKIntegrator": ("mesh.integration", "RKIntegrator"),
         "RuntimeParameters": ("util.runparams", "RuntimeParameters"),
         "TimerCollection": ("util.profile_pyro", "TimerCollection")}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    raise AttributeError(name)
