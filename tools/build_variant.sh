#!/bin/bash
# developer tool: a second library that differs from the product build in the flags of SOME units
# (for A/B runs on the GPU box through PYRO2_AMD_LIB); the other objects are the product build's.
#   tools/build_variant.sh w1 "-DPYRO_ADVM_WPE3=1" advection adv_fast
# -> pyro2_amd/lib/libpyrohip_w1.so
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/.."
python - "$name" "$flags" "$@" <<'PY'
import os, subprocess, sys
sys.path.insert(0, ".")
from pyro2_amd import build as b
name, flags, units = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
b.build()
objs = []
for src, obj, extra in b.units():
    o = os.path.join(b.LIBDIR, "obj", obj + ".o")
    if obj in units:
        o = os.path.join(b.LIBDIR, "obj", f"{obj}_{name}.o")
        subprocess.check_call([b._hipcc(), f"--offload-arch={b.ARCH}"] + b.COMMON + extra + flags +
                              ["-c", os.path.join(b.CSRC, src), "-o", o])
    objs.append(o)
lib = os.path.join(b.LIBDIR, f"libpyrohip_{name}.so")
subprocess.check_call([b._hipcc(), f"--offload-arch={b.ARCH}", "-shared", "-o", lib] + objs +
                      ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
print(lib)
PY
