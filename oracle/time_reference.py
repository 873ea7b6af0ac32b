"""Time THE REFERENCE ITSELF (pyro2, unmodified) on this host's cores for the two
legs it runs at its true speed here: the advection step (pure NumPy) and the
multigrid V-cycle (pure NumPy).  The compressible step is njit code and numba is
not installable here, so its timing is the identity-njit
interpreter speed and is reported as such (small grid only) next to the C port's.

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Run in the build container (the
reference does not exist on the GPU box):

    cd /tmp && MPLBACKEND=Agg \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference \
      /opt/conda/bin/python3.9 /root/repo/oracle/time_reference.py [--quick]

writes profiles/cpu_reference.json (host, core count, date, per-leg rates); bench.py
reads that file for the `cpu_baseline` objects of its advection / multigrid legs
(kind "reference", measured in the build container, not on the GPU box -- the file
says which host).  Follows /root/reference/pyro/advection/simulation.py:56-94 (evolve),
pyro/pyro_sim.py:241-281 (single_step) and pyro/multigrid/MG.py:623-697 (solve).
"""
import json
import os
import platform
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "profiles", "cpu_reference.json")
os.chdir(tempfile.mkdtemp())      # Pyro writes inputs.auto into cwd

import pyro.multigrid.MG as MG                      # noqa: E402
from pyro.pyro_sim import Pyro                      # noqa: E402


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def time_advection(nx, steps, warmup=1):
    """Pyro("advection") smooth, nx^2 periodic, limiter 2: seconds per Pyro.single_step
    (ghost fill + compute_timestep + evolve), BASELINE configs[1] at nx = 2048"""
    p = Pyro("advection")
    p.initialize_problem("smooth", inputs_dict={"mesh.nx": nx, "mesh.ny": nx,
                                                "driver.max_steps": 10 ** 6,
                                                "driver.tmax": 1.0e9,
                                                "particles.do_particles": 0})
    for _ in range(warmup):
        p.single_step()
    t0 = time.perf_counter()
    for _ in range(steps):
        p.single_step()
    el = time.perf_counter() - t0
    return {"workload": f"advection smooth {nx}x{nx} periodic, limiter 2 (Pyro.single_step of the "
                        "unmodified reference, pure NumPy)",
            "nx": nx, "steps": steps, "seconds_per_step": el / steps,
            "value": nx * nx * steps / el, "unit": "cell-updates/s", "cores": 1}


def mg_rhs(x, y):
    return -2.0 * ((1.0 - 6.0 * x ** 2) * y ** 2 * (1.0 - y ** 2) +
                   (1.0 - 6.0 * y ** 2) * x ** 2 * (1.0 - x ** 2))


def time_mg(nx, cycles):
    """CellCenterMG2d(nx, nx) all-Dirichlet Poisson of multigrid/examples/mg_test_simple.py:
    seconds per V-cycle inside solve() (BASELINE configs[3] at nx = 4096).  solve(rtol=1e-30)
    never converges, so exactly max_cycles V-cycles (with their norms) are timed."""
    a = MG.CellCenterMG2d(nx, nx, xl_BC_type="dirichlet", yl_BC_type="dirichlet",
                          xr_BC_type="dirichlet", yr_BC_type="dirichlet", verbose=0)
    a.init_zeros()
    a.init_RHS(mg_rhs(a.x2d, a.y2d))
    a.max_cycles = cycles
    t0 = time.perf_counter()
    a.solve(rtol=1.0e-30)
    el = time.perf_counter() - t0
    return {"workload": f"multigrid constant-coeff Poisson {nx}x{nx} dirichlet, {a.num_cycles} V-cycles "
                        "(nsmooth 10, bottom 50) of the unmodified reference's solve(), pure NumPy",
            "nx": nx, "cycles": int(a.num_cycles), "seconds_per_vcycle": el / a.num_cycles,
            "value": a.num_cycles / el, "unit": "V-cycles/s", "cores": 1,
            "residual_error_after": float(a.residual_error)}


def time_compressible(nx, steps):
    """identity-njit speed of the compressible step (NOT numba speed; labelled)"""
    p = Pyro("compressible")
    p.initialize_problem("sedov", inputs_dict={"mesh.nx": nx, "mesh.ny": nx,
                                               "driver.max_steps": 10 ** 6})
    p.single_step()
    t0 = time.perf_counter()
    for _ in range(steps):
        p.single_step()
    el = time.perf_counter() - t0
    return {"workload": f"compressible sedov {nx}x{nx} (reference with numba.njit replaced by the "
                        "identity: interpreter speed of the njit kernels, NOT what numba delivers)",
            "nx": nx, "steps": steps, "seconds_per_step": el / steps,
            "value": nx * nx * steps / el, "unit": "cell-updates/s", "cores": 1}


def time_compressible_numpy_stages(nx, steps):
    """The reference's compressible step WITHOUT its njit kernels: interface.states,
    riemann_hllc and artificial_viscosity (pyro/compressible/interface.py:5,239,
    riemann.py:681 -- the code numba compiles) are replaced by stubs that only allocate their
    results, so what is timed is every NumPy stage of unsplit_fluxes.py:134-549 and
    simulation.py:290-450 (primitive variables, flattening, limiting, the conversions, the
    transverse corrections, sources, the update) plus the driver's ghost fill and time step.
    The real reference also runs the three kernels: its step is LONGER than this, its rate
    BELOW the one returned here.  The stub flux is zero, so the state stays the physical
    initial state for every timed step."""
    import pyro.compressible.interface as ifc
    import pyro.compressible.riemann as riemann
    saved = (ifc.states, ifc.artificial_viscosity, riemann.riemann_hllc)

    def states(idir, ng, dx, dloga, dt, irho, iu, iv, ip, ix, nspec, gamma, qv, dqv):
        return np.zeros_like(qv), np.zeros_like(qv)

    def avisc(ng, dx, dy, Lx, Ly, xmin, ymin, coord_type, cvisc, u, v):
        return np.zeros_like(u), np.zeros_like(u)

    def hllc(idir, ng, idens, ixmom, iymom, iener, irhoX, nspec, lower_solid, upper_solid,
             gamma, U_l, U_r):
        return np.zeros_like(U_l)
    ifc.states, ifc.artificial_viscosity, riemann.riemann_hllc = states, avisc, hllc
    try:
        p = Pyro("compressible")
        p.initialize_problem("sedov", inputs_dict={"mesh.nx": nx, "mesh.ny": nx,
                                                   "driver.max_steps": 10 ** 6})
        p.single_step()
        t0 = time.perf_counter()
        for _ in range(steps):
            p.single_step()
        el = time.perf_counter() - t0
    finally:
        ifc.states, ifc.artificial_viscosity, riemann.riemann_hllc = saved
    return {"workload": f"compressible sedov {nx}x{nx}: Pyro.single_step of the reference with its three "
                        "njit kernels (interface.states, riemann_hllc, artificial_viscosity) replaced by "
                        "allocate-only stubs -- the NumPy stages alone; the real step takes longer",
            "nx": nx, "steps": steps, "seconds_per_step": el / steps,
            "value_upper_bound": nx * nx * steps / el, "unit": "cell-updates/s", "cores": 1}


def time_diffusion(nx, steps):
    """Pyro("diffusion") gaussian nx^2 (pure NumPy + the multigrid solver: the reference at its
    true speed): seconds per Pyro.single_step, pyro/diffusion/simulation.py:70-122"""
    p = Pyro("diffusion")
    p.initialize_problem("gaussian", inputs_dict={"mesh.nx": nx, "mesh.ny": nx,
                                                  "driver.max_steps": 10 ** 6, "driver.tmax": 1.0e9})
    p.single_step()
    t0 = time.perf_counter()
    for _ in range(steps):
        p.single_step()
    el = time.perf_counter() - t0
    return {"workload": f"diffusion gaussian {nx}x{nx} (inputs.gaussian), Pyro.single_step",
            "nx": nx, "steps": steps, "seconds_per_step": el / steps,
            "value": nx * nx * steps / el, "unit": "cell-updates/s", "cores": 1}


def only_diffusion():
    """add / refresh the diffusion section of profiles/cpu_reference.json (the other legs keep
    their numbers and date; this section carries its own)"""
    out = json.load(open(OUT))
    sec = {"date": time.strftime("%Y-%m-%d %H:%M:%S %Z")}
    for nx, steps in ((512, 3), (2048, 2)):
        r = time_diffusion(nx, steps)
        sec[str(nx)] = r
        print("diffusion", nx, r["seconds_per_step"], "s/step", r["value"], "cells/s", flush=True)
    out["diffusion"] = sec
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.abspath(OUT))


def time_solver_identity_njit(solver, problem, nx, ny, steps, inputs_file=None, extra=None):
    """a solver whose step is mostly njit code, at the speed of the identity-njit shim (the
    INTERPRETER's speed: a lower bound of what numba delivers, labelled as such)"""
    p = Pyro(solver)
    d = {"mesh.nx": nx, "mesh.ny": ny, "driver.max_steps": 10 ** 6, "driver.tmax": 1.0e9}
    d.update(extra or {})
    p.initialize_problem(problem, inputs_file=inputs_file, inputs_dict=d)
    p.single_step()
    t0 = time.perf_counter()
    for _ in range(steps):
        p.single_step()
    el = time.perf_counter() - t0
    return {"workload": f"{solver} {problem} {nx}x{ny}" + (f" ({inputs_file})" if inputs_file else "") +
                        ": Pyro.single_step of the reference with numba.njit replaced by the identity -- "
                        "interpreter speed of the njit kernels, NOT what numba delivers",
            "nx": nx, "ny": ny, "steps": steps, "seconds_per_step": el / steps,
            "value": nx * ny * steps / el, "unit": "cell-updates/s", "cores": 1}


def only_f4():
    """SURVEY 8(f4) solvers (swe, compressible_rk, SphericalPolar compressible): sections of
    their own in profiles/cpu_reference.json.  Their steps are njit code, so
    the reference itself can only be timed at interpreter speed here (small grids, labelled);
    the C port's rate at the bench size is recorded beside it by oracle/gen_fullsize.py's log."""
    out = json.load(open(OUT))
    date = time.strftime("%Y-%m-%d %H:%M:%S %Z")
    for key, args in (("swe", ("swe", "dam", 128, 10, 2, "inputs.dam.x")),
                      ("compressible_rk", ("compressible_rk", "sedov", 64, 64, 1, None)),
                      ("compressible_spherical", ("compressible", "sedov", 64, 64, 1, "inputs.sedov.spherical"))):
        r = time_solver_identity_njit(*args)
        out[key] = {"date": date, "interpreted": r}
        print(key, r["seconds_per_step"], "s/step", r["value"], "cells/s", flush=True)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.abspath(OUT))


def main():
    if "--only-diffusion" in sys.argv:
        return only_diffusion()
    if "--only-f4" in sys.argv:
        return only_f4()
    quick = "--quick" in sys.argv
    out = {"host": platform.node(), "cpu": cpu_model(), "host_cores": os.cpu_count(),
           "cores_used": 1,
           "date": time.strftime("%Y-%m-%d %H:%M:%S %Z"),
           "python": sys.version.split()[0], "numpy": np.__version__,
           "note": "the reference is single-threaded NumPy (+ numba njit without parallel); measured "
                   "in the build container (the GPU box has no /root/reference), conda python3.9, "
                   "numba.njit = identity shim (irrelevant for advection / multigrid: no njit code "
                   "on those paths)",
           "advection": {}, "multigrid": {}, "compressible_identity_njit": {},
           "compressible_numpy_stages_only": {}}
    for nx, steps in ((512, 4), (2048, 3)) if not quick else ((256, 2),):
        r = time_advection(nx, steps)
        out["advection"][str(nx)] = r
        print("advection", nx, r["seconds_per_step"], "s/step", r["value"], "cells/s", flush=True)
    for nx, cyc in ((1024, 3), (4096, 2)) if not quick else ((256, 2),):
        r = time_mg(nx, cyc)
        out["multigrid"][str(nx)] = r
        print("multigrid", nx, r["seconds_per_vcycle"], "s/V-cycle", flush=True)
    for nx, steps in ((64, 2),):
        r = time_compressible(nx, steps)
        out["compressible_identity_njit"][str(nx)] = r
        print("compressible", nx, r["seconds_per_step"], "s/step", flush=True)
    for nx, steps in ((512, 4), (1024, 3), (2048, 2)) if not quick else ((128, 2),):
        r = time_compressible_numpy_stages(nx, steps)
        out["compressible_numpy_stages_only"][str(nx)] = r
        print("compressible, NumPy stages only", nx, r["seconds_per_step"], "s/step",
              r["value_upper_bound"], "cells/s at most", flush=True)
    if not quick:
        with open(OUT, "w") as f:
            json.dump(out, f, indent=1)
        print("wrote", os.path.abspath(OUT))
    else:
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
