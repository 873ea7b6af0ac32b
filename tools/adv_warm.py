"""advection leg: untimed steps in front, then five timed legs in a row (cold clock? stable afterwards?)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from pyro2_amd import device
ctx = device.Context(0)
for nx, steps, warm in ((2048, 600, 3000), (8192, 60, 300)):
    for rep in range(5):
        r = bench.bench_advection(ctx, device, nx=nx, steps=steps, warmup=warm if rep == 0 else 6, fast_math=1, other=False)
        print(nx, "rep", rep, "ms/step", r["ms_per_step"], "frac", r.get("roofline", {}).get("step_frac"))
