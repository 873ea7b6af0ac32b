#!/bin/bash
# developer tool: multigrid variants in ONE gpurun call (box-to-box variance ~5 %)
mkdir -p gpurun_out
SIZES="${@:-512 2048 4096}"
{
PYRO_MG_BAND_R=8 python -m pytest tests/test_device_multigrid.py tests/test_incompressible.py -m gpu -x -q 2>&1 | tail -2
for R in 4 8 4 8; do
echo "=== band kernel rows per wave $R"
PYRO_MG_BAND_R=$R python tools/mg_prof.py $SIZES | grep -E "nx=|smooth"
done
} > gpurun_out/mg_ab.log 2>&1
tail -80 gpurun_out/mg_ab.log
