"""developer tool: phase times inside the band smoother kernel (pyrohip_mg_tuning.trace), one V-cycle"""
import os, sys
TRACE_TUNING = dict(trace=1)   # pass to DeviceMG(..., tuning=TRACE_TUNING)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyro2_amd import device
ctx = device.Context(0)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = (np.arange(nx + 2) - 0.5) / nx
X, Y = np.meshgrid(x, x, indexing="ij")
m = device.DeviceMG(ctx, nx, tuning=TRACE_TUNING)
L = m.nlevels - 1
m.zero(L, 0); m.set(L, 1, np.sin(X) * Y); m.init_rhs_norm()
m.solve(rtol=0.0, max_cycles=2)
ctx.sync()
