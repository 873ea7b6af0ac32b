import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from pyro2_amd import device
ctx = device.Context(0)
nx = 64
rng = np.random.default_rng(0)
rhs = rng.standard_normal((nx + 2, nx + 2))
for ns, nb in ((0, 0), (0, 200), (10, 50), (40, 0)):
    m = device.DeviceMG(ctx, nx, nsmooth=ns, nsmooth_bottom=nb)
    L = m.nlevels - 1
    m.zero(L, 0); m.set(L, 1, rhs); m.init_rhs_norm()
    for _ in range(3): m.vcycle()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(50): m.vcycle()
    ctx.sync()
    t1 = time.perf_counter()
    sweeps = 2 * (ns * 2 * 5 + nb)
    print(f"nsmooth={ns} bottom={nb}: {(t1-t0)/50*1e6:.1f} us per V-cycle (launch only, no solve loop), {sweeps} colour sweeps", flush=True)
