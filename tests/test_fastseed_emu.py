"""The CONTRACTED builds (gpu.fast_math = 1: what bench.py times) on the CPU suite's emulator WITH
THE PRECISION OF THE GPU FORMS.  The product's fast units take reciprocals and
roots from v_rcp / v_rsq seeds refined by one Newton step (~3e-14 relative), intrinsics the host
does not have: the default emulator build gives those units true divisions, so their ALGEBRA is
covered on the CPU but not its behaviour under inexact quotients.  The variant built here
(PYRO_EMU_DEFS=-DPYRO_EMU_FASTSEED: 2^-23 seeds + one Newton / Goldschmidt step, csrc/hydro.h)
has the GPU forms' precision; every contracted kernel is held to north_star's tolerance against
the oracle on it: 1e-10 element-wise (compressible, compressible_rk, swe), 1e-12 (advection).
Runs in a subprocess: the variant is a second library."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, os, sys
import numpy as np
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")]
import build_emu
from oracle import orc
from pyro2_amd import _lib, device
lib = build_emu.build()
assert lib.endswith("_emu_build_fastseed/libpyrohip_emu.so"), lib
_lib.use_library(lib, allow_backends=("host-emu",))
ctx = device.Context(0)
from conftest import comp_floors, elementwise_err
from helpers import RK_TABLEAU, DtPolicy, meta_to_params, oracle_comp_run, oracle_rk_step
from sedov_ic import sedov_ic
out = {}

# the variant really has inexact quotients: a contracted step differs from the default emulator's
# bit-faithful one in the last digits
nx = 64
ic, meta, bcs = sedov_ic(nx, r_init=0.1)
Uo, dto, _ = oracle_comp_run(ic, meta, bcs, 0.1, 8)
fl = comp_floors(Uo[4:-4, 4:-4])
for ks in (1, 2):
    P = device.make_comp_params(meta[3], meta[4], kernel_set=ks, fast_math=1, march_rows=13 if ks == 2 else 0)
    s = device.DeviceState(ctx, nx, nx, 4, [list(r) for r in orc.comp_var_bcs(bcs)])
    s.upload(np.nan_to_num(ic))
    pol, dts = DtPolicy(0.1), []
    for _ in range(8):
        s.fill_bc()
        dt = pol(s.comp_dt(P, 0.8))
        s.comp_step(P, dt)
        pol.advance(dt)
        dts.append(dt)
    U = s.download()
    out[f"comp_ks{ks}"] = max(elementwise_err(U[4:-4, 4:-4, n], Uo[4:-4, 4:-4, n], fl[n]) for n in range(4))
    out[f"comp_ks{ks}_dt"] = float(np.abs(np.array(dts) / dto - 1).max())
    out[f"comp_ks{ks}_differs"] = bool((U[4:-4, 4:-4] != Uo[4:-4, 4:-4]).any())
    # ... and the same steps enqueued on the device
    s.upload(np.nan_to_num(ic))
    pol = DtPolicy(0.1)
    s.comp_evolve(P, 0.8, pol, 8)
    U = s.download()
    out[f"comp_ks{ks}_evolve"] = max(elementwise_err(U[4:-4, 4:-4, n], Uo[4:-4, 4:-4, n], fl[n]) for n in range(4))

# compressible_rk: one RK4 step in four launches, contracted
Po, cfl = meta_to_params(meta, bcs)
Ur = np.nan_to_num(ic.copy())
orc.comp_fill_bc(Ur, nx, nx, 4, bcs, Po.gamma, Po.grav, Po.dy)
dt = 0.01 * orc.comp_rk_dt(Ur, nx, nx, 4, Po.dx, Po.dy, Po.gamma, cfl)
for _ in range(3):
    oracle_rk_step(Ur, Po, bcs, dt, "RK4")
a, b = RK_TABLEAU["RK4"]
P = device.make_comp_params(meta[3], meta[4], kernel_set=2, fast_math=1, march_rows=13)
s = device.DeviceState(ctx, nx, nx, 4, [list(r) for r in orc.comp_var_bcs(bcs)])
k = device.DeviceState(ctx, nx, nx, 4, [["outflow"] * 4] * 16)
s.upload(np.nan_to_num(ic))
for _ in range(3):
    s.comp_rk_step(P, k, dt, a, b)
U = s.download()
flr = comp_floors(Ur[4:-4, 4:-4])
out["rk4"] = max(elementwise_err(U[4:-4, 4:-4, n], Ur[4:-4, 4:-4, n], flr[n]) for n in range(4))

# advection, contracted, three steps per launch
x = (np.arange(nx + 8) - 3.5) / nx
X, Y = np.meshgrid(x, x, indexing="ij")
a0 = 1.0 + np.exp(-60.0 * ((X - 0.5) ** 2 + (Y - 0.5) ** 2))
dta = orc.adv_dt(1 / nx, 1 / nx, 1.0, 1.0, 0.8)
ao = a0.copy()
for _ in range(9):
    orc.fill_ghost(ao, nx, nx, 4, ("periodic",) * 4)
    orc.adv_step(ao, nx, nx, 4, 1 / nx, 1 / nx, 1.0, 1.0, dta, 2)
sa = device.DeviceState(ctx, nx, nx, 4, [["periodic"] * 4])
sa.upload(a0)
sa.adv_evolve(0, 1 / nx, 1 / nx, 1.0, 1.0, [dta] * 9, 2, fast_math=1)
out["adv"] = float(np.abs(sa.download()[4:-4, 4:-4, 0] - ao[4:-4, 4:-4]).max() / np.abs(ao).max())

# shallow water: 5 steps of the contracted one-launch kernel, Roe and HLLC
from fullsize_ics import SWE_BCS, swe_dam2d_ic, swe_meta
m = swe_meta(nx, nx)
vb = orc.comp_var_bcs(SWE_BCS)
rows = [list(vb[0]), list(vb[2]), list(vb[3]), list(vb[0])]
for rs in ("Roe", "HLLC"):
    Pw = orc.swe_params(nx, nx, 4, m[3], m[4], m[5], int(m[6]), rs)
    Uw = swe_dam2d_ic(nx)
    icw = Uw.copy()
    dts = []
    for _ in range(5):
        for n in range(4):
            orc.fill_ghost(Uw, nx, nx, 4, rows[n], n=n)
        dts.append(0.5 * orc.swe_dt(Uw, Pw, m[7]))
        orc.swe_step(Uw, Pw, dts[-1])
    s = device.DeviceState(ctx, nx, nx, 4, rows)
    s.upload(icw)
    for dt in dts:
        s.fill_bc()
        s.swe_step(m[3], m[4], m[5], int(m[6]), rs, dt, fast_math=1)
    U = s.download()[4:-4, 4:-4]
    ref = Uw[4:-4, 4:-4]
    out["swe_" + rs] = float(max(np.abs(U[..., n] - ref[..., n]).max() / max(np.abs(ref[..., n]).max(), 1e-300)
                                 for n in range(3)))
print("RESULT " + json.dumps(out))
'''


def test_contracted_builds_with_gpu_precision_quotients_on_the_emulator(tmp_path):
    env = dict(os.environ, PYRO_EMU_NAME="fastseed", PYRO_EMU_DEFS="-DPYRO_EMU_FASTSEED",
               PYRO_EMU_FAST_ONLY="1", PYRO_TEST_WORKERS="0")
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build()             # (the default build: the variant links its bit-faithful objects)
    script = tmp_path / "fastseed.py"
    script.write_text(SCRIPT)
    p = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, env=env,
                       timeout=1500, cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    print(res)
    for k in ("comp_ks1", "comp_ks2", "comp_ks1_evolve", "comp_ks2_evolve", "rk4", "swe_Roe", "swe_HLLC"):
        assert res[k] <= 1e-10, (k, res)
    assert res["comp_ks1_dt"] <= 1e-10 and res["comp_ks2_dt"] <= 1e-10
    assert res["adv"] <= 1e-12, res
    # the variant is not the exact-division emulator in disguise
    assert res["comp_ks1_differs"] and res["comp_ks2_differs"]
