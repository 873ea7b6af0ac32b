"""developer tool: time per compressible step on small grids (device-side stepping, Sedov)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from pyro2_amd import device
from pyro2_amd.decomp import DtPolicy
from sedov_ic import sedov_ic
ctx = device.Context(0)
for nx in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024]:
    ic, meta, bcs = sedov_ic(nx, r_init=0.05)
    s = device.DeviceState(ctx, nx, nx, 4, [["outflow"] * 4] * 4)
    s.upload(np.nan_to_num(ic))
    P = device.make_comp_params(1.0 / nx, 1.0 / nx, fast_math=1, kernel_set=-1)
    pol = DtPolicy(1e9)
    s.comp_evolve(P, 0.8, pol, 50)
    ctx.sync()
    n = 400
    t0 = time.perf_counter()
    s.comp_evolve(P, 0.8, pol, n)
    ctx.sync()
    t1 = time.perf_counter()
    print(f"nx={nx:5d}: {(t1 - t0) / n * 1e6:8.1f} us per step  ({nx * nx / ((t1 - t0) / n) / 1e9:6.2f} Gcell/s)")
