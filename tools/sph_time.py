"""developer tool (GPU box): SphericalPolar Sedov through Pyro.run_sim(), the staged kernel set
(gpu.kernel_set = 0: nine launches per step) against the one-launch tile kernel (default)"""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R)
import bench
from pyro2_amd import device
ctx = device.Context(0)
for nx in (int(a) for a in os.environ.get("SIZES", "512,2048").split(",")):
    for ks in (0, -1):
        r = bench.bench_pyro_run(ctx, device, "compressible", "sedov",
                                 {"mesh.nx": nx, "mesh.ny": nx, "gpu.kernel_set": ks}, 10, 2,
                                 inputs_file="inputs.sedov.spherical")
        print(f"spherical sedov {nx}^2 kernel_set {ks}: {r['ms_per_step']:.4f} ms/step  {r['value'] / 1e9:.2f} Gcell/s", flush=True)
