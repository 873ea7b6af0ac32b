"""developer probe: element-wise errors of the fast build (run on the GPU box from the repo root)"""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from conftest import elementwise_err, comp_floors, max_rel_err, GOLDEN
import test_device_compressible as T
from pyro2_amd import device
from helpers import oracle_comp_run
from sedov_ic import sedov_ic
ctx = device.Context(0)
class D: kind = "hip"
dev = ctx; dev.kind = "hip"
def g(name): return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
ic, meta, bcs = sedov_ic(512)
Uo, dto, _ = oracle_comp_run(ic, meta, bcs, 0.1, 30)
for ks in (1, 2):
    U, dts, _ = T.device_comp_run(dev, ic, meta, bcs, 0.1, 30, fast_math=1, kernel_set=ks)
    fl = comp_floors(Uo[4:-4, 4:-4])
    print("sedov512 kset", ks, "elementwise", [elementwise_err(U[4:-4, 4:-4, n], Uo[4:-4, 4:-4, n], fl[n]) for n in range(4)],
          "arraywide", [max_rel_err(U[4:-4, 4:-4, n], Uo[4:-4, 4:-4, n]) for n in range(4)])
for name, tm, cap in (("comp_quad_0606", None, 1000), ("comp_rt_0945", None, 10000)):
    gg = g(name); bc = [str(b) for b in gg["bc"]]
    for fast in (0, 1):
        U, dts, t = T.device_comp_run(dev, gg["ic"], gg["meta"], bc, float(gg["tmax"]), cap, fast_math=fast, kernel_set=2)
        ref = gg["gold"]; I = U[4:-4, 4:-4]
        fl = [float(np.median(np.abs(ref[..., n]))) or float(np.abs(ref[..., n]).max()) for n in range(4)]
        print(name, "fast", fast, "steps", len(dts), "elementwise(median floor)", [elementwise_err(I[..., n], ref[..., n], fl[n]) for n in range(4)],
              "arraywide", [max_rel_err(I[..., n], ref[..., n]) for n in range(4)])
