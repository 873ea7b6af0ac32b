"""developer tool: advection step timing by size and strip length (GPU box)"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
from pyro2_amd import device
ctx = device.Context(0)
# SIZES="2048:0,12,13;8192:32,64" (rows 0 = the library's choice)
SPEC = os.environ.get("SIZES", "2048:0;8192:0")
for nx, rows_list in [(int(a.split(":")[0]), a.split(":")[1]) for a in SPEC.split(";")]:
    x = (np.arange(nx + 8) - 3.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    ic = 1.0 + np.exp(-60.0 * ((X - 0.5) ** 2 + (Y - 0.5) ** 2))
    for rows in rows_list.split(","):
      for fast in (1, 0) if os.environ.get("BOTH", "0") == "1" else (1,):
        st = device.DeviceState(ctx, nx, nx, 4, [["periodic"] * 4])
        st.upload(ic)
        dt = 0.8 / nx
        for fused in (1,):
            def step():
                if not fused:
                    st.fill_bc()
                st.adv_step(0, 1 / nx, 1 / nx, 1.0, 1.0, dt, int(os.environ.get("LIM", "2")), fill=bool(fused), fast_math=fast,
                            march_rows=int(rows))
            for _ in range(10): step()
            noprof = os.environ.get("NOPROF", "0") == "1"
            ctx.sync(); ctx.prof_enable(not noprof)
            n = 200 if nx <= 2048 else 50
            t0 = time.perf_counter()
            for _ in range(n): step()
            ctx.sync(); t1 = time.perf_counter()
            if noprof:
                print(f"nx={nx} rows={rows} fast={fast} step {1e6*(t1-t0)/n:8.2f} us (no event timers)", flush=True)
                continue
            prof = ctx.prof_report(); ctx.prof_enable(False)
            k, ms = prof["k_adv_step"]
            print(f"nx={nx} rows={rows} fast={fast} fused={fused} step {1e6*(t1-t0)/n:8.1f} us  kernel {1e3*ms/k:8.1f} us  "
                  f"{16*nx*nx/(ms/k*1e-3)/1e12:.2f} TB/s kernel, {16*nx*nx*n/(t1-t0)/1e12:.2f} TB/s step")
