"""one secondary bench leg alone (to be run under rocprofv3: tools/profile_round.sh,
tools/pmc_also.sh, tools/pmc_leg.sh)

    python tools/also_run.py adv|mg|swe|rk|sph|diff      (NX, FM, STEPS from the environment)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyro2_amd import device    # noqa: E402
import bench                    # noqa: E402

ctx = device.Context(0)
what = sys.argv[1]
FM = int(os.environ.get("FM", "1"))
STEPS = int(os.environ.get("STEPS", "10"))


def nx(default):
    return int(os.environ.get("NX", str(default)))


if what == "adv":
    r = bench.bench_advection(ctx, device, nx=nx(2048), steps=60, warmup=6, fast_math=FM, other=False)
    print(r["ms_per_step"], r["roofline"]["kernel_avg_ms"])
elif what == "mg":
    r = bench.bench_mg(ctx, device, nx=nx(4096), cycles=10, small_sizes=False)
    print(r["ms_per_vcycle"])
else:
    n = nx({"swe": 4096, "rk": 4096, "sph": 2048, "diff": 2048}[what])
    size = {"mesh.nx": n, "mesh.ny": n, "gpu.fast_math": FM}
    if what == "swe":
        r = bench.bench_pyro_run(ctx, device, "swe", "dam", size, STEPS, 3, inputs_file="inputs.dam.x")
    elif what == "rk":
        r = bench.bench_pyro_run(ctx, device, "compressible_rk", "sedov", size, STEPS, 3)
    elif what == "sph":
        r = bench.bench_pyro_run(ctx, device, "compressible", "sedov", size, STEPS, 2,
                                 inputs_file="inputs.sedov.spherical")
    else:
        r = bench.bench_pyro_run(ctx, device, "diffusion", "gaussian", size, STEPS, 2)
    print(what, n, r["ms_per_step"], "ms/step", r["value"] / 1e9, "Gcell/s")
