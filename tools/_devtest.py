import sys, os, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from pyro2_amd import device
from sedov_ic import sedov_ic
from helpers import DtPolicy
from test_device_compressible import comp_state
g=np.load('tests/golden/comp_sedov_1024_developed.npz')
hip=device.Context(0)
nx, ng, nsteps = 1024, 4, int(g["nsteps"])
ic, meta, bcs = sedov_ic(nx)
for ks in (1, 2):
    s = comp_state(hip, nx, nx, bcs); s.upload(ic)
    P = device.make_comp_params(1.0 / nx, 1.0 / nx, fast_math=1, kernel_set=ks)
    pol = DtPolicy(0.1); dts=[]
    while pol.t < 0.1 and pol.n < nsteps + 10:
        dts.extend(s.comp_evolve(P, 0.8, pol, min(256, nsteps + 10 - pol.n)))
    I = s.download()[ng:-ng, ng:-ng]
    patch = I[nx // 2 - 8:nx // 2 + 8, nx // 2:]
    ref = g["patch"]
    from conftest import comp_floors
    fl = comp_floors(ref)
    errs=[float((np.abs(patch[...,v]-ref[...,v])/(np.abs(ref[...,v])+fl[v])).max()) for v in range(4)]
    step = nx // 64
    S=I[::step, ::step]; R=g["samples"]
    fl2 = comp_floors(R)
    errl=[float((np.abs(S[...,v]-R[...,v])/(np.abs(R[...,v])+fl2[v])).max()) for v in range(4)]
    print(os.environ.get('PYRO2_AMD_LIB','current'), 'ks', ks, 'len', len(dts), 'patch', ['%.2e'%e for e in errs], 'lattice', ['%.2e'%e for e in errl], 'dt err', float(np.abs(np.array(dts[:-1])/g['dts'][:-1]-1).max()))
