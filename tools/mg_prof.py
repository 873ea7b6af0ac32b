"""developer tool: per-kernel time inside V-cycles at one size (HIP-event profiling of the library)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyro2_amd import device
ctx = device.Context(0)
for nx in [int(a) for a in sys.argv[1:]] or [2048]:
    x = (np.arange(nx + 2) - 0.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    rhs = -2.0 * ((1 - 6 * X**2) * Y**2 * (1 - Y**2) + (1 - 6 * Y**2) * X**2 * (1 - X**2))
    bc = os.environ.get("PYRO_MG_PROF_BC")
    m = device.DeviceMG(ctx, nx, bcs=tuple(bc.split(',')) if ',' in bc else (bc,) * 4) if bc else device.DeviceMG(ctx, nx)
    L = m.nlevels - 1
    m.zero(L, 0); m.set(L, 1, rhs); m.init_rhs_norm()
    m.solve(rtol=0.0, max_cycles=2); m.zero(L, 0); ctx.sync()
    t0 = time.perf_counter(); m.solve(rtol=0.0, max_cycles=10); ctx.sync(); t1 = time.perf_counter()
    ctx.prof_enable(True); m.solve(rtol=0.0, max_cycles=10); prof = ctx.prof_report(); ctx.prof_enable(False)
    tot = sum(ms for _, ms in prof.values())
    print(f"nx={nx}: {(t1-t0)/10*1e6:.0f} us per V-cycle wall; sum of kernel events {tot/10*1e3:.0f} us")
    for k, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:28s} {n/10:6.1f} launches/cycle  {ms/n*1e3:8.1f} us each  {ms/10*1e3:8.1f} us per cycle")
