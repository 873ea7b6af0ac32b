"""developer tool: per-kernel time of one incompressible step (shear, 2048^2) + wall time of its parts"""
import sys, os, time, io, contextlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyro2_amd import device
from pyro2_amd.pyro_sim import Pyro
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = device.Context(0)
device.Context._default = ctx
os.chdir(tempfile.mkdtemp())
with contextlib.redirect_stdout(io.StringIO()):
    p = Pyro("incompressible")
    p.initialize_problem("shear", inputs_dict={"mesh.nx": nx, "mesh.ny": nx, "driver.max_steps": 10})
    p.single_step(); p.single_step()
ctx.sync()
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(3):
        p.single_step()
ctx.sync()
t1 = time.perf_counter()
print(f"nx={nx}: {(t1-t0)/3*1e3:.2f} ms per step, V-cycles per step {p.sim.mg_cycles}")
ctx.prof_enable(True)
with contextlib.redirect_stdout(io.StringIO()):
    p.single_step()
prof = ctx.prof_report(); ctx.prof_enable(False)
tot = sum(ms for _, ms in prof.values())
print(f"sum of kernel events {tot:.2f} ms")
for k, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"   {k:30s} {n:5d} launches {ms/n*1e3:9.1f} us each {ms:8.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    p.single_step()
ctx.sync(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
