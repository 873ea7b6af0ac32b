"""multigrid V-cycle leg with different untimed V-cycle counts in front (is the short leg on a cold clock?)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from pyro2_amd import device
ctx = device.Context(0)
for nx in (4096, 2048, 512):
    x = (np.arange(nx + 2) - 0.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    rhs = -2.0 * ((1.0 - 6.0 * X ** 2) * Y ** 2 * (1.0 - Y ** 2) + (1.0 - 6.0 * Y ** 2) * X ** 2 * (1.0 - X ** 2))
    m = device.DeviceMG(ctx, nx)
    L = m.nlevels - 1
    m.zero(L, 0); m.set(L, 1, rhs); m.init_rhs_norm()
    for warm in (2, 20, 80, 200, 200):
        m.zero(L, 0)
        m.solve(rtol=0.0, max_cycles=warm)
        m.zero(L, 0)
        ctx.sync()
        t0 = time.perf_counter()
        m.solve(rtol=0.0, max_cycles=10)
        ctx.sync()
        t1 = time.perf_counter()
        print(nx, "warm", warm, "ms/V-cycle", (t1 - t0) / 10 * 1e3)
        time.sleep(0.5) if warm == 200 else None
