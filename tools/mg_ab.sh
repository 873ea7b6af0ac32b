#!/bin/bash
# developer tool: multigrid variants in ONE gpurun call (box-to-box variance ~5 %)
mkdir -p gpurun_out
SIZES="${@:-512 2048 4096}"
{
python -m pytest tests/test_device_multigrid.py tests/test_incompressible.py tests/test_diffusion.py tests/test_host_api.py -m gpu -x -q 2>&1 | tail -3
echo "=== defaults"
python tools/mg_prof.py $SIZES
python tools/mgc_trace.py 2>&1 | tail -1
for WT in -1 2 3; do
echo "=== coarse kernel wave_top=$WT"
PYRO_MGC_WAVE_TOP=$WT python tools/mg_prof.py 512 | grep -E "nx=|coarse"
PYRO_MGC_WAVE_TOP=$WT python tools/mgc_trace.py 2>&1 | tail -1
done
} > gpurun_out/mg_ab.log 2>&1
tail -80 gpurun_out/mg_ab.log
