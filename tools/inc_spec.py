"""developer tool: the incompressible bench leg alone (PYRO_MG_SPEC_DEBUG=1 prints every solve's cycles)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyro2_amd import device
import bench
ctx = device.Context(0)
print(bench.bench_incompressible(ctx, device, nx=int(sys.argv[1]) if len(sys.argv) > 1 else 2048, steps=3))
