#!/bin/bash
# developer tool: A/B of multigrid variants in ONE gpurun call (box-to-box variance ~5 %)
mkdir -p gpurun_out
{
python -m pytest tests/test_device_multigrid.py tests/test_incompressible.py tests/test_diffusion.py -m gpu -x -q 2>&1 | tail -3
for WT in -1 1 2 3; do
echo "=== PYRO_MG_WAVE=0 coarse wave_top=$WT"
PYRO_MG_WAVE=0 PYRO_MGC_WAVE_TOP=$WT python tools/mg_prof.py 512 | grep -E "nx=|coarse"
done
echo "=== no pow2"
PYRO_MG_WAVE=0 PYRO_MG_NOPOW2=1 python tools/mg_prof.py 512 | grep -E "nx=|coarse"
} > gpurun_out/mg_ab.log 2>&1
tail -80 gpurun_out/mg_ab.log
