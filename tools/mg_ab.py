#!/usr/bin/env python3
"""A/B timing of the multigrid smoother variants on the GPU (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyro2_amd import device

ctx = device.Context(0)
for nx in (4096,):
    x = (np.arange(nx + 2) - 0.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    rhs = -2.0 * ((1 - 6 * X**2) * Y**2 * (1 - Y**2) + (1 - 6 * Y**2) * X**2 * (1 - X**2))
    for kind in [int(k) for k in os.environ.get('MG_KINDS', '23,13').split(',')]:
        m = device.DeviceMG(ctx, nx)
        m.set_smoother(kind)
        L = m.nlevels - 1
        m.zero(L, 0); m.set(L, 1, rhs); m.init_rhs_norm()
        m.solve(rtol=0.0, max_cycles=2)
        m.zero(L, 0)
        ctx.sync()
        t0 = time.perf_counter()
        nc, res, rel = m.solve(rtol=0.0, max_cycles=10)
        ctx.sync(); t1 = time.perf_counter()
        ctx.prof_enable(True)
        m.solve(rtol=0.0, max_cycles=10)
        prof = ctx.prof_report(); ctx.prof_enable(False)
        print(f"nx={nx} smoother={kind}: {(t1-t0)/10*1e3:.3f} ms/V-cycle, {10/(t1-t0):.1f} V-cycles/s, res={res:.3e}",
              {k: (n, round(ms / 10, 3)) for k, (n, ms) in prof.items()}, flush=True)
        del m
