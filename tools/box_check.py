"""is this GPU box one of the slow ones (a box of rounds 3 / 4 ran memory-bound kernels at half
speed)?  One advection step at 8192^2: ~275 us on a normal box.  Exit code 1 when slow."""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R)
import numpy as np
from pyro2_amd import device
ctx = device.Context(0)
nx = 8192
st = device.DeviceState(ctx, nx, nx, 4, [["periodic"] * 4])
st.upload(np.ones((nx + 8, nx + 8)))
for _ in range(3):
    st.adv_step(0, 1 / nx, 1 / nx, 1.0, 1.0, 0.8 / nx, 2, fill=True, fast_math=1)
ctx.sync(); t0 = time.perf_counter()
for _ in range(10):
    st.adv_step(0, 1 / nx, 1 / nx, 1.0, 1.0, 0.8 / nx, 2, fill=True, fast_math=1)
ctx.sync(); us = (time.perf_counter() - t0) / 10 * 1e6
print(f"box check: advection 8192^2 step {us:.0f} us ({'SLOW box' if us > 400 else 'normal'})")
# ... and a multigrid V-cycle at 4096^2: ~0.71 ms on a normal box (boxes that pass the advection
# check have run it at 1.4 ms)
n = 4096
x = (np.arange(n + 2) - 0.5) / n
X, Y = np.meshgrid(x, x, indexing="ij")
m = device.DeviceMG(ctx, n)
L = m.nlevels - 1
m.zero(L, 0); m.set(L, 1, -2.0 * ((1 - 6 * X**2) * Y**2 * (1 - Y**2) + (1 - 6 * Y**2) * X**2 * (1 - X**2))); m.init_rhs_norm()
m.solve(rtol=0.0, max_cycles=2); m.zero(L, 0); ctx.sync()
t0 = time.perf_counter(); m.solve(rtol=0.0, max_cycles=6); ctx.sync()
mg_us = (time.perf_counter() - t0) / 6 * 1e6
print(f"box check: multigrid 4096^2 V-cycle {mg_us:.0f} us ({'SLOW box' if mg_us > 950 else 'normal'})")
sys.exit(1 if (us > 400 or mg_us > 950) else 0)
