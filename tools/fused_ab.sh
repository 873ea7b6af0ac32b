#!/bin/bash
# A/B timing of compiled variants of the library on the GPU box (developer tool):
#   bash tools/fused_ab.sh libbase.so libpyrohip.so ...
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
NX=${NX:-8192}
for lib in "$@"; do
  for fm in 1 0; do
    PYRO2_AMD_LIB=$R/pyro2_amd/lib/$lib timeout 300 python bench.py --nx $NX --steps 20 --warmup 5 --no-also --no-cpu-baseline --fast-math $fm 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib fm=$fm nx=$NX', round(d['roofline']['update_kernels_ms_per_step'],4), 'ms kernel,', round(d['ms_per_step'],4), 'ms/step')"
  done
done | tee $O/fused_ab.log
