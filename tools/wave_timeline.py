"""per-wavefront lives of one k_ctu_wave launch (developer build: tools/build_variant.sh tl "-DPYRO_WAVE_TIMELINE" wave_fast;
PYRO2_AMD_LIB=pyro2_amd/lib/libpyrohip_tl.so python tools/wave_timeline.py [nx] [steps])

Prints, on the 100 MHz constant clock: the launch's span, the distribution of wavefront lives and of their end times,
and the means by XCD, by SIMD slot, by row strip and by column-strip position -- where a one-round launch loses the
time between the average wavefront's life and the kernel's duration."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyro2_amd import _lib, device                                   # noqa: E402
from pyro2_amd.compressible.problems.sedov import sedov_state        # noqa: E402
from pyro2_amd.decomp import DtPolicy                                # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = device.Context(0)
st = device.DeviceState(ctx, nx, nx, 4, [["outflow"] * 4] * 4)
st.upload(sedov_state(nx, nx, 4, 0.0, 1.0, 0.0, 1.0, 1.4, 0.01, 4))
P = device.make_comp_params(1.0 / nx, 1.0 / nx, fast_math=1, kernel_set=-1)
pol = DtPolicy(1.0e9)
st.comp_evolve(P, 0.8, pol, steps)
ctx.sync()
lib = _lib.lib()
ncb = (nx + 55) // 56
fn = lib.pyrohip_debug_wave_timeline
fn.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(4 * 65536, dtype=np.uint64)
assert fn(buf.ctypes.data, 65536) == 0
T = buf.reshape(-1, 4)
T = T[T[:, 1] > 0]
t0, t1 = T[:, 0].astype(np.int64), T[:, 1].astype(np.int64)
hw = (T[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
xcc = (T[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
sb = (T[:, 3] >> np.uint64(32)).astype(np.int64)
cb = (T[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
base = t0.min()
life = (t1 - t0) * 0.01          # us
start = (t0 - base) * 0.01
end = (t1 - base) * 0.01
print(f"{len(T)} wavefronts, launch span {end.max():.1f} us; start: mean {start.mean():.1f} max {start.max():.1f}; "
      f"life: mean {life.mean():.1f} min {life.min():.1f} max {life.max():.1f}; end: mean {end.mean():.1f} "
      f"p50 {np.percentile(end, 50):.1f} p90 {np.percentile(end, 90):.1f} p99 {np.percentile(end, 99):.1f}")
wave_id, simd_id, cu_id, sh_id, se_id = hw & 0xf, (hw >> 4) & 0x3, (hw >> 8) & 0xf, (hw >> 12) & 0x1, (hw >> 13) & 0x7


def by(name, key):
    ks = np.unique(key)
    if len(ks) > 40:
        ks = ks[:: max(1, len(ks) // 20)]
    print(name + ": " + "  ".join(f"{k}:{life[key == k].mean():.0f}/{end[key == k].mean():.0f}" for k in ks))


span = end.max()
print("wavefronts alive at span - t: " + "  ".join(
    f"{t:.0f} us: {int(np.sum((start <= span - t) & (end > span - t)))}" for t in (800, 600, 400, 300, 200, 150, 100, 50, 20)
    if t < span))
print(f"wavefront-time lost to the drain (slots not held between the last dispatch and the end): "
      f"{np.sum(span - end[end > start.max()]) / len(np.unique(hw | (xcc << 20))) :.1f} us per SIMD slot pair")
print("(mean life / mean end time, us)")
by("xcc", xcc)
by("wave slot", wave_id)
by("simd", simd_id)
by("se", se_id)
by("cu", cu_id)
by("row strip", sb)
by("column strip", cb)
# the two wavefronts of a SIMD: who ends first, by how much
key = (xcc * 8 + se_id) * 64 + cu_id * 4 + simd_id
d = {}
for k, e in zip(key, end):
    d.setdefault(int(k), []).append(e)
pairs = np.array([sorted(v)[-2:] for v in d.values() if len(v) >= 2])
print(f"SIMDs with >= 2 wavefronts: {len(pairs)}; the later one ends {np.mean(pairs[:, 1] - pairs[:, 0]):.1f} us after the "
      f"earlier one on average (max {np.max(pairs[:, 1] - pairs[:, 0]):.1f}); SIMD end times: mean {pairs[:, 1].mean():.1f} "
      f"p10 {np.percentile(pairs[:, 1], 10):.1f} p90 {np.percentile(pairs[:, 1], 90):.1f} max {pairs[:, 1].max():.1f}")
