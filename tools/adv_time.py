#!/usr/bin/env python3
"""time of one advection step at 2048^2 (developer tool)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pyro2_amd import device  # noqa: E402

ctx = device.Context(0)
nx = int(os.environ.get("ADV_NX", "2048"))
s = device.DeviceState(ctx, nx, nx, 4, [[3, 3, 3, 3]])
x = (np.arange(nx + 8) - 3.5) / nx
X, Y = np.meshgrid(x, x, indexing="ij")
s.upload(np.ascontiguousarray((1.0 + np.exp(-60.0 * ((X - 0.5)**2 + (Y - 0.5)**2)))[:, :, None]))
dx = 1.0 / nx
dt = 0.8 * dx
for _ in range(20):
    s.fill_bc(); s.adv_step(0, dx, dx, 1.0, 1.0, dt, 2)
ctx.sync()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    s.fill_bc(); s.adv_step(0, dx, dx, 1.0, 1.0, dt, 2)
ctx.sync()
t1 = time.perf_counter()
ctx.prof_enable(True)
for _ in range(20):
    s.fill_bc(); s.adv_step(0, dx, dx, 1.0, 1.0, dt, 2)
prof = ctx.prof_report(); ctx.prof_enable(False)
print(f"nx={nx}: {(t1 - t0) / n * 1e6:.1f} us/step = {nx * nx / ((t1 - t0) / n) / 1e9:.1f} Gcell/s",
      {k: round(ms / cnt * 1e3, 1) for k, (cnt, ms) in prof.items()}, "sum", float(s.download().sum()))
