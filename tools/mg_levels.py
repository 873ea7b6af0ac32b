#!/usr/bin/env python3
"""per-level cost of the multigrid building blocks (developer tool)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyro2_amd import device
ctx = device.Context(0)
nx = 4096
for kind in (0, 12, 13):
    m = device.DeviceMG(ctx, nx); m.set_smoother(kind)
    print("smoother", kind)
    for lev in range(m.nlevels):
        n = 2 ** (lev + 1)
        m.smooth(lev, 10); ctx.sync()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps): m.smooth(lev, 10)
        ctx.sync(); ts = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps): m.residual(lev)
        ctx.sync(); tr = (time.perf_counter() - t0) / reps
        tp = trs = 0
        if lev > 0:
            t0 = time.perf_counter()
            for _ in range(reps): m.restrict(lev)
            ctx.sync(); trs = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps): m.prolong_add(lev)
            ctx.sync(); tp = (time.perf_counter() - t0) / reps
        print(f"  n={n:5d} smooth(10) {ts*1e6:9.1f} us  residual {tr*1e6:7.1f}  restrict {trs*1e6:7.1f}  prolong {tp*1e6:7.1f}", flush=True)
