"""developer tool (GPU box): compressible step time by grid size and strip length of the
row-marching kernel.  SIZES="4096:0,37,76,152;8192:0,128,149" (0 = the library's choice); SL="3,1": step_launches values to
run (3 = boundary fill, dt policy and step kernel as three launches per step; 1 = one); CHECK=1
compares dt sequence and end state (crc) of the variants of a size; PROF=1: per-kernel durations"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
from pyro2_amd import device
from pyro2_amd.compressible.problems.sedov import sedov_state
from pyro2_amd.decomp import DtPolicy
ctx = device.Context(0)
SPEC = os.environ.get("SIZES", "4096:0;8192:0")
FM = int(os.environ.get("FM", "1"))
for nx, rows_list in [(int(a.split(":")[0]), a.split(":")[1]) for a in SPEC.split(";")]:
    ic = None
    for rows, sl in [(r, int(q)) for r in rows_list.split(",") for q in os.environ.get("SL", "0").split(",")]:
        st = device.DeviceState(ctx, nx, nx, 4, [["outflow"] * 4] * 4)
        for r0 in range(0, nx + 8, 512):
            nr = min(512, nx + 8 - r0)
            blk = sedov_state(nx, nx, 4, 0.0, 1.0, 0.0, 1.0, 1.4, 0.01, 4, i0=r0, ni=nr)
            if os.environ.get("AMBIENT") == "1":       # no blast: every cell the ambient gas
                blk = np.broadcast_to(np.nan_to_num(sedov_state(nx, nx, 4, 0.0, 1.0, 0.0, 1.0, 1.4, 0.01, 4, i0=8, ni=1))[0, 8], blk.shape).copy()
            st.upload_rows(r0, blk)
        P = device.make_comp_params(1.0 / nx, 1.0 / nx, fast_math=FM, kernel_set=int(os.environ.get("KS", "2")), march_rows=int(rows),
                                    step_launches=sl)
        pol = DtPolicy(1.0e9)
        dts = list(st.comp_evolve(P, 0.8, pol, 5))
        ctx.sync()
        n = int(os.environ.get("STEPS", "0")) or max(10, int(2.0e9 / (nx * nx)))
        t0 = time.perf_counter()
        dts += list(st.comp_evolve(P, 0.8, pol, n))
        ctx.sync()
        ms = (time.perf_counter() - t0) / n * 1e3
        if os.environ.get("PROF") == "1":      # per-kernel durations (HIP events around each launch)
            ctx.prof_enable(True)
            st.comp_evolve(P, 0.8, pol, 20)
            ctx.sync()
            print("   ", {k: (v[0], round(1e3 * v[1] / v[0], 1)) for k, v in ctx.prof_report().items()}, "(launches, us each)")
            ctx.prof_enable(False)
        line = f"nx={nx} march_rows={rows} fm={FM} step_launches={sl}: {ms:8.3f} ms/step  {nx * nx / ms / 1e6:6.2f} Gcell/s"
        if os.environ.get("CHECK") == "1" and nx <= 8192:     # dt sequence and end state of the variants of a size
            import zlib
            U = st.download()
            sig = (tuple(dts), zlib.crc32(U.tobytes()))
            if ic is None or ic[0] != (nx, rows):
                ic = ((nx, rows), sig)
                line += "  (reference of the check)"
            else:
                line += "  IDENTICAL dts + state" if sig == ic[1] else f"  DIFFERENT (dts equal: {sig[0] == ic[1][0]})"
            del U
        print(line, flush=True)
        del st
