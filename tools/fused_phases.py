#!/usr/bin/env python3
"""time of k_ctu_fused truncated after phase k (libraries built with
-DPYRO_FUSED_STOP=k, see tools/fused_phases.sh) -- developer tool"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

from pyro2_amd import device  # noqa: E402
from pyro2_amd.compressible.problems.sedov import sedov_state  # noqa: E402

nx = int(os.environ.get("NX", "4096"))
ctx = device.Context(0)
P = device.make_comp_params(dx=1.0 / nx, dy=1.0 / nx, fast_math=1, kernel_set=1)
s = device.DeviceState(ctx, nx, nx, 4, [["outflow"] * 4] * 4)
ic = sedov_state(nx, nx, 4, 0.0, 1.0, 0.0, 1.0, 1.4, 0.01, 4, i0=0, ni=nx + 8)
s.upload_rows(0, ic)
s.fill_bc()
dt = 1e-7
ms = []
for it in range(6):
    ctx.prof_enable(True)
    try:
        s.comp_step(P, dt)
    except Exception as e:      # truncated kernels leave no valid CFL number
        pass
    prof = ctx.prof_report()
    ctx.prof_enable(False)
    if it:
        ms.append(prof["k_ctu_fused"][1] / prof["k_ctu_fused"][0])
    s.upload_rows(0, ic)
    s.fill_bc()
print(os.environ.get("PYRO2_AMD_LIB", "default").split("/")[-1], f"nx={nx}", "k_ctu_fused ms:", round(float(np.median(ms)), 4))
