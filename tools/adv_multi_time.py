"""developer tool: advection, several steps per launch (pyrohip_adv_evolve) -- time per STEP
by grid size, steps per launch and chunk length (GPU box).

  SPEC="2048:1/0,2/0,2/19;8192:2/0,3/0"   size:K/rows,...   (K = 0: the single-step kernel,
                                          rows = 0: the library's choice)
  UV="-1,-1" (advection velocity), PRIO=0/1, FAST=1/0, LIM=2, CHECK=1 (compare the end state with single steps, bit for bit
  in the exact build)
"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
from pyro2_amd import device

ctx = device.Context(0)
SPEC = os.environ.get("SPEC", "2048:0/0,1/0,2/0,3/0;8192:0/0,2/0,3/0")
PRIO = int(os.environ.get("PRIO", "0"))
FAST = int(os.environ.get("FAST", "1"))
LIM = int(os.environ.get("LIM", "2"))
CHECK = int(os.environ.get("CHECK", "0"))
UU, VV = (float(a) for a in os.environ.get("UV", "1,1").split(","))      # advection velocity
for part in SPEC.split(";"):
    nx = int(part.split(":")[0])
    x = (np.arange(nx + 8) - 3.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    ic = 1.0 + np.exp(-60.0 * ((X - 0.5) ** 2 + (Y - 0.5) ** 2))
    dt = 0.8 / nx
    ref = None
    for kr in part.split(":")[1].split(","):
        K, rows = (int(v) for v in kr.split("/"))
        st = device.DeviceState(ctx, nx, nx, 4, [["periodic"] * 4])
        st.upload(ic)
        nsteps = 12 * (60 if nx <= 2048 else (20 if nx <= 4096 else 6))

        def run(n):
            if K == 0:
                for _ in range(n):
                    st.adv_step(0, 1 / nx, 1 / nx, UU, VV, dt, LIM, fill=True, fast_math=FAST, march_rows=rows)
            else:
                st.adv_evolve(0, 1 / nx, 1 / nx, UU, VV, [dt] * n, LIM, fast_math=FAST, march_rows=rows,
                              multi_k=K, multi_prio=PRIO)
        run(12)
        ctx.sync()
        t0 = time.perf_counter()
        run(nsteps)
        ctx.sync()
        t1 = time.perf_counter()
        us = 1e6 * (t1 - t0) / nsteps
        line = f"nx={nx} K={K} rows={rows} fast={FAST} prio={PRIO}: {us:8.2f} us per step, {16 * nx * nx / us / 1e6:.2f} TB/s " \
               f"= {16 * nx * nx / us / 1e6 / 8:.3f} of 8 TB/s, {nx * nx / us / 1e3:.1f} Gcell/s"
        if CHECK:
            a = st.download()[:, :, 0]
            if ref is None:
                ref = a
                line += "  (reference of the check)"
            else:
                d = np.abs(a - ref).max()
                line += f"  max diff vs first variant {d:.3e}" + (" IDENTICAL" if np.array_equal(a, ref) else "")
        print(line, flush=True)
