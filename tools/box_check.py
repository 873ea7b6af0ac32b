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
sys.exit(1 if us > 400 else 0)
