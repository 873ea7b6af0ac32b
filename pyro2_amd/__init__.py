"""pyro2_amd -- the per-timestep hot path of python-hydro/pyro2 on AMD
Instinct MI355X (gfx950): hand-written HIP kernels behind pyro's own Python
class surface (Pyro / Simulation / CellCenterData2d / Grid2d, MG.CellCenterMG2d).

    from pyro2_amd import Pyro
    p = Pyro("compressible"); p.initialize_problem("sedov"); p.run_sim()

There is no CPU fallback: importing works anywhere, the first device operation
needs pyro2_amd/lib/libpyrohip.so (python -m pyro2_amd.build) and a GPU.
"""
__version__ = "0.1.0"


def __getattr__(name):
    if name == "Pyro":
        from .pyro_sim import Pyro
        return Pyro
    raise AttributeError(name)
