"""Build libpyrohip.so (HIP, gfx950) in-tree with hipcc.

`python -m pyro2_amd.build` or __graft_entry__.build().  hipcc cross-compiles
without a GPU.  The library is written to pyro2_amd/lib/libpyrohip.so so that
it travels with the source snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, os.environ.get("PYRO_LIB_NAME", "libpyrohip.so"))
EXTRA = os.environ.get("PYRO_EXTRA_FLAGS", "").split()   # developer experiments only
FAST_EXTRA = os.environ.get("PYRO_FAST_EXTRA_FLAGS", "").split()   # ... fast_math units only

# scheduling strategy of the row-marching compressible kernel (developer experiments:
# PYRO_WAVE_SCHED=default builds it with the compiler's own choice)
_WS = os.environ.get("PYRO_WAVE_SCHED", "max-ilp")
WAVE_SCHED = [] if _WS == "default" else ["-mllvm", f"-amdgpu-sched-strategy={_WS}"]

ARCH = "gfx950"
LAST_BUILD = None     # what the last build() call did: {"mode": "reused" | "compiled", ...} (__graft_entry__ prints it)
COMMON = ["-std=c++17", "-fPIC", "-O3"]

# (source, object name, extra flags).  The compressible kernels are built
# twice: bit-faithful (no FMA contraction) and contracted.
UNITS = [
    ("ctx.hip", "ctx", ["-ffp-contract=off"]),
    ("advection.hip", "advection", ["-ffp-contract=off", "-DPYRO_FAST=0"]),
    ("advection.hip", "adv_fast", ["-ffp-contract=fast", "-DPYRO_FAST=1"]),
    ("compressible.hip", "comp_exact", ["-ffp-contract=off", "-DPYRO_FAST=0"]),
    ("compressible.hip", "comp_fast", ["-ffp-contract=fast", "-DPYRO_FAST=1"]),
    ("comp_fused.hip", "fused_exact", ["-ffp-contract=off", "-DPYRO_FAST=0"]),
    ("comp_fused.hip", "fused_fast", ["-ffp-contract=fast", "-DPYRO_FAST=1"]),
    # max-ilp scheduling: the row-marching kernel runs two wavefronts per SIMD, what
    # hides latency there is independent work inside a wavefront (12.30 -> 12.15 ms)
    ("comp_wave.hip", "wave_exact", ["-ffp-contract=off", "-DPYRO_FAST=0"] + WAVE_SCHED),
    # (contracted build: -fno-honor-nans drops the v_max x, x canonicalisations in front of every
    # fmin / fmax of a loaded or lane-moved value, 22 per row; a valid state has no NaN, an invalid
    # one is caught by the positivity flag)
    ("comp_wave.hip", "wave_fast", ["-ffp-contract=fast", "-DPYRO_FAST=1", "-fno-honor-nans"] + WAVE_SCHED),
    # the row-marching kernel of SphericalPolar grids (round 6)
    ("comp_sph_wave.hip", "sphw_exact", ["-ffp-contract=off", "-DPYRO_FAST=0"]),
    # (no -fno-honor-nans here: ghost faces of an oddly reflected density hold NaN roots that the
    # fmin / fmax floors of the CGF solver are there to absorb)
    ("comp_sph_wave.hip", "sphw_fast", ["-ffp-contract=fast", "-DPYRO_FAST=1"]),
    ("comp_api.hip", "comp_api", ["-ffp-contract=off"]),
    ("multigrid.hip", "multigrid", ["-ffp-contract=off"]),
    ("mg_march.hip", "mg_march", ["-ffp-contract=off", "-mllvm", "-pragma-unroll-threshold=200000"]),
    # ... its instances with a tail (MGMarch::tail) in units of their own: minutes each
    ("mg_march.hip", "mg_march_t1", ["-ffp-contract=off", "-mllvm", "-pragma-unroll-threshold=200000", "-DMGM_UNIT=1"]),
    ("mg_march.hip", "mg_march_t2", ["-ffp-contract=off", "-mllvm", "-pragma-unroll-threshold=200000", "-DMGM_UNIT=2"]),
    ("incompressible.hip", "incompressible", ["-ffp-contract=off"]),
    ("swe.hip", "swe", ["-ffp-contract=off", "-DPYRO_FAST=0"]),
    ("swe.hip", "swe_fast", ["-ffp-contract=fast", "-DPYRO_FAST=1", "-fno-honor-nans"]),
    ("comm.hip", "comm", ["-ffp-contract=off"]),
]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
        [os.path.join(HERE, "..", "include", "pyrohip.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def units():
    us = [u for u in UNITS if os.path.exists(os.path.join(CSRC, u[0]))]
    return us


def build(force=False, verbose=False):
    os.makedirs(os.path.join(LIBDIR, "obj"), exist_ok=True)
    hipcc = _hipcc()
    deps = _deps()
    us = units()
    flagline = " ".join(EXTRA + FAST_EXTRA + WAVE_SCHED)
    libstamp = LIB + ".flags"
    global LAST_BUILD
    if not force and not _stale(LIB, deps) and os.path.exists(libstamp) and \
            open(libstamp).read() == flagline:
        LAST_BUILD = {"mode": "reused", "compiled_units": [], "library": LIB,
                      "why": "the library in the tree is newer than every source and was built with these flags"}
        return LIB
    compiled = []

    def compile_one(u):
        src, name, extra = u
        obj = os.path.join(LIBDIR, "obj", name + os.environ.get("PYRO_OBJ_SUFFIX", "") + ".o")
        fx = FAST_EXTRA if "-DPYRO_FAST=1" in extra else []
        cmd = [hipcc, f"--offload-arch={ARCH}"] + COMMON + extra + EXTRA + fx + \
              ["-c", os.path.join(CSRC, src), "-o", obj]
        # an object newer than its source and every header is kept -- if it was compiled by
        # exactly this command line (the flags are recorded beside it: an object left behind
        # by a build with experiment flags is never linked into a flag-free library)
        own = [d for d in deps if not d.endswith(".hip")] + [os.path.join(CSRC, src)]
        stamp, line = obj + ".flags", " ".join(cmd)
        same = os.path.exists(stamp) and open(stamp).read() == line
        if not force and same and not _stale(obj, own + [os.path.abspath(__file__)]):
            return obj
        if verbose:
            print(line)
        subprocess.check_call(cmd)
        compiled.append(name)
        with open(stamp, "w") as f:
            f.write(line)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(us))) as ex:
        objs = list(ex.map(compile_one, us))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-o", LIB] + objs
    if any(u[1] == "comm" for u in us):
        cmd += ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(libstamp, "w") as f:
        f.write(flagline)
    LAST_BUILD = {"mode": "compiled", "compiled_units": sorted(compiled), "units": len(us), "library": LIB}
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
