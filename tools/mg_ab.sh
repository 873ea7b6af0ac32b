#!/bin/bash
# developer tool: multigrid variants in ONE gpurun call (box-to-box variance ~5 %)
mkdir -p gpurun_out
SIZES="${@:-512 2048 4096}"
{
python -m pytest tests/test_device_multigrid.py tests/test_incompressible.py tests/test_diffusion.py -m gpu -x -q 2>&1 | tail -3
for BM in 1024 2048 4096; do
echo "=== band kernel up to $BM^2"
PYRO_MG_BAND=$BM python tools/mg_prof.py $SIZES | grep -E "nx=|smooth"
done
echo "=== band up to 4096, 10 iterations per launch up to 1024"
PYRO_MG_BAND=4096 PYRO_MG_NSMALL=1024 python tools/mg_prof.py $SIZES | grep -E "nx=|smooth"
python tools/mg_trace.py 2048 2>&1 | grep "n=2048\|n=1024" | head -4
} > gpurun_out/mg_ab.log 2>&1
tail -80 gpurun_out/mg_ab.log
