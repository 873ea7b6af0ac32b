"""developer tool: phase times inside the coarse V-cycle kernel (PYRO_MGC_TRACE=1)"""
import os, sys
os.environ["PYRO_MGC_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyro2_amd import device
ctx = device.Context(0)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = (np.arange(nx + 2) - 0.5) / nx
X, Y = np.meshgrid(x, x, indexing="ij")
m = device.DeviceMG(ctx, nx)
L = m.nlevels - 1
m.zero(L, 0); m.set(L, 1, np.sin(X) * Y); m.init_rhs_norm()
m.solve(rtol=0.0, max_cycles=3)
ctx.sync()
