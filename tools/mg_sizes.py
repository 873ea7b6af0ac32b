#!/usr/bin/env python3
"""V-cycle time of the constant-coefficient multigrid solver at several grid
sizes (developer tool): the coarse levels (<= 64^2, one LDS-resident workgroup)
dominate on the grids the incompressible / diffusion problems use."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pyro2_amd import device  # noqa: E402

ctx = device.Context(0)
for nx in [int(a) for a in os.environ.get("MG_SIZES", "64,128,256,512,1024,2048,4096").split(",")]:
    x = (np.arange(nx + 2) - 0.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    rhs = -2.0 * ((1 - 6 * X**2) * Y**2 * (1 - Y**2) + (1 - 6 * Y**2) * X**2 * (1 - X**2))
    m = device.DeviceMG(ctx, nx, tuning=dict(march_tail=int(os.environ.get('MG_TAIL', '1')), coarse_wave=int(os.environ.get('MG_WAVE', '1')), march_waves=int(os.environ.get('MG_WAVES', '0')), march_side=float(os.environ.get('MG_SIDE', '1.5')), march_minrows=int(os.environ.get('MG_MINROWS', '32')), nsmall=int(os.environ.get('MG_NSMALL', '512'))))
    L = m.nlevels - 1
    m.zero(L, 0); m.set(L, 1, rhs); m.init_rhs_norm()
    m.solve(rtol=0.0, max_cycles=3)
    m.zero(L, 0)
    ctx.sync()
    ncyc = 20
    t0 = time.perf_counter()
    nc, res, rel = m.solve(rtol=0.0, max_cycles=ncyc)
    ctx.sync()
    t1 = time.perf_counter()
    print(f"nx={nx}: {(t1 - t0) / ncyc * 1e6:.1f} us/V-cycle  res={res:.6e}", flush=True)
    del m
