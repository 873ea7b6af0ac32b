"""the advection / multigrid bench legs alone (to be run under rocprofv3, tools/pmc_also.sh)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyro2_amd import device
import bench
ctx = device.Context(0)
what = sys.argv[1]
if what == "adv":
    r = bench.bench_advection(ctx, device, nx=int(os.environ.get("NX", "2048")), steps=60, warmup=6,
                              fast_math=int(os.environ.get("FM", "1")), other=False)
    print(r["ms_per_step"], r["roofline"]["kernel_avg_ms"])
else:
    r = bench.bench_mg(ctx, device, nx=int(os.environ.get("NX", "4096")), cycles=10, small_sizes=False)
    print(r["ms_per_vcycle"])
