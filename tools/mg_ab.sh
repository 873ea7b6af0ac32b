#!/bin/bash
# developer tool: multigrid variants in ONE gpurun call (box-to-box variance ~5 %)
mkdir -p gpurun_out
SIZES="${@:-512 2048 4096}"
{
python -m pytest tests/test_device_multigrid.py tests/test_incompressible.py tests/test_diffusion.py tests/test_host_api.py -m gpu -x -q 2>&1 | tail -3
echo "=== defaults"
python tools/mg_prof.py $SIZES
python tools/mgc_trace.py 2>&1 | tail -1
echo "=== round-1 smoother (LDS tile kernel, 5 iterations per launch, 9-operation update)"
PYRO_MG_BAND=0 PYRO_MG_KSMALL=5 PYRO_MG_NOPOW2=1 HIP_FORCE_DEV_KERNARG=0 python tools/mg_prof.py $SIZES | grep -E "nx="
} > gpurun_out/mg_ab.log 2>&1
tail -80 gpurun_out/mg_ab.log
