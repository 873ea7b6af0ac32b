#!/bin/bash
# phase ablation of k_ctu_fused (developer tool): builds libstop<k>.so with
# -DPYRO_FUSED_STOP=k (CPU side, before gpurun), times them on the GPU box
if [ "$1" = "build" ]; then
  for k in 0 1 2 3 4; do PYRO_LIB_NAME=libstop$k.so PYRO_OBJ_SUFFIX=_stop$k PYRO_FAST_EXTRA_FLAGS="-DPYRO_FUSED_STOP=$k" python -m pyro2_amd.build >/dev/null 2>&1 & done; wait
else
  for l in libstop0.so libstop1.so libstop2.so libstop3.so libstop4.so libpyrohip.so; do PYRO2_AMD_LIB=$PWD/pyro2_amd/lib/$l python tools/fused_phases.py 2>&1 | tail -1; done
fi
