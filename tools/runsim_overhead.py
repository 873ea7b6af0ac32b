"""where a short Pyro.run_sim() leg spends its time: the C call against the whole run_sim (developer tool)
    python tools/runsim_overhead.py swe|rk|sph [steps]"""
import contextlib
import io
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyro2_amd import device          # noqa: E402
from pyro2_amd.pyro_sim import Pyro   # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "swe"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = {"swe": ("swe", "dam", "inputs.dam.x", 4096, "swe_evolve"),
       "rk": ("compressible_rk", "sedov", None, int(os.environ.get("NX", "4096")), "comp_rk_evolve"),
       "sph": ("compressible", "sedov", "inputs.sedov.spherical", 2048, "comp_evolve")}[what]
ctx = device.Context(0)
device.Context._default = ctx
os.chdir(tempfile.mkdtemp())
inner = []
orig = getattr(device.DeviceState, cfg[4])


def timed(self, *a, **k):
    t0 = time.perf_counter()
    r = orig(self, *a, **k)
    inner.append(time.perf_counter() - t0)
    return r


setattr(device.DeviceState, cfg[4], timed)
with contextlib.redirect_stdout(io.StringIO()):
    p = Pyro(cfg[0])
    p.initialize_problem(cfg[1], inputs_file=cfg[2],
                         inputs_dict={"mesh.nx": cfg[3], "mesh.ny": cfg[3], "gpu.fast_math": 1,
                                      "driver.max_steps": 3, "driver.tmax": 1.0e9})
    p.run_sim()
    ctx.sync()
    res = []
    for rep in range(3):
        inner.clear()
        p.sim.max_steps = p.sim.n + steps
        t0 = time.perf_counter()
        p.run_sim()
        ctx.sync()
        t1 = time.perf_counter()
        res.append((t1 - t0, sum(inner), len(inner)))
for tot, inn, n in res:
    print(f"{what} {steps} steps: run_sim {tot * 1e3:.3f} ms ({tot / steps * 1e3:.4f} per step), inside the C call "
          f"{inn * 1e3:.3f} ms in {n} call(s), host around it {1e3 * (tot - inn):.3f} ms")
