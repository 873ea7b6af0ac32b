"""Build the host-emulated libpyrohip (TEST INFRASTRUCTURE ONLY).

Compiles the *same* pyro2_amd/csrc/*.hip kernel sources with g++ against
tests/emu/hip/hip_runtime.h into tests/_emu_build/libpyrohip_emu.so, so the
GPU-less build container can execute the kernels (slowly) and compare them
with the oracle before GPU time is spent.  The library identifies itself as
backend "host-emu"; pyro2_amd._lib refuses it unless a test injects it.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# PYRO_EMU_NAME / PYRO_EMU_DEFS: a second emulated library built with extra -D flags (e.g. the
# marching smoother with several strips per workgroup: PYRO_EMU_NAME=g4 PYRO_EMU_DEFS=-DMGM_G=4)
_NAME = os.environ.get("PYRO_EMU_NAME", "")
OUT = os.path.join(ROOT, "tests", "_emu_build" + ("_" + _NAME if _NAME else ""))
LIB = os.path.join(OUT, "libpyrohip_emu.so")

sys.path.insert(0, ROOT)
from pyro2_amd import build as hb  # noqa: E402


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    deps = hb._deps() + [os.path.join(HERE, "hipemu.cpp"), os.path.join(HERE, "comm_emu.cpp"),
                         os.path.join(HERE, "hip", "hip_runtime.h")]
    if not force and not hb._stale(LIB, deps):
        return LIB
    flags = ["-std=c++17", "-O1", "-g", "-fPIC", "-ffp-contract=off",
             "-I" + HERE, "-DPYRO_EMU=1", '-DPYRO_BACKEND_NAME="host-emu"'] + \
        os.environ.get("PYRO_EMU_DEFS", "").split()
    units = [u for u in hb.units() if u[1] not in ("comm",)]

    headers = [d for d in deps if not d.endswith(".hip")] + [os.path.abspath(__file__)]

    # a variant built with extra -D flags for the CONTRACTED units only (PYRO_EMU_FAST_ONLY=1, e.g.
    # -DPYRO_EMU_FASTSEED, which only the PYRO_FAST code reads) links the other units' objects of
    # the default build instead of compiling them again
    fast_only = bool(_NAME) and os.environ.get("PYRO_EMU_FAST_ONLY") == "1"
    base_out = os.path.join(ROOT, "tests", "_emu_build")

    def cc(u):
        src, name, extra = u
        defs = [f for f in extra if f.startswith("-D")]
        obj = os.path.join(OUT, name + ".o")
        if fast_only and "-DPYRO_FAST=1" not in extra:
            base = os.path.join(base_out, name + ".o")
            if os.path.exists(base) and not hb._stale(base, headers + [os.path.join(hb.CSRC, src)]):
                return base
        # an object newer than its own source and every header is kept
        if not force and not hb._stale(obj, headers + [os.path.join(hb.CSRC, src)]):
            return obj
        subprocess.check_call(["g++"] + flags + defs + ["-x", "c++", "-c",
                              os.path.join(hb.CSRC, src), "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, units))
    rt = os.path.join(OUT, "hipemu.o")
    subprocess.check_call(["g++"] + flags + ["-c", os.path.join(HERE, "hipemu.cpp"), "-o", rt])
    cm = os.path.join(OUT, "comm_emu.o")
    subprocess.check_call(["g++"] + flags + ["-c", os.path.join(HERE, "comm_emu.cpp"), "-o", cm])
    subprocess.check_call(["g++", "-shared", "-o", LIB] + objs + [rt, cm])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
