#!/bin/bash
# scratch: one GPU-box session (edited per use)
cd "$GRAFT_REPO_ROOT"
TAG=r03y TESTS=1 BENCH=1 ALSOSTATS=1 bash tools/gpu_r03.sh
TAG=r03y bash tools/pmc_also.sh > gpurun_out/pmc_also_r03y.log 2>&1; tail -30 gpurun_out/pmc_also_r03y.log
python tools/mg_sizes.py > gpurun_out/r03y_mg_sizes.txt 2>&1; cat gpurun_out/r03y_mg_sizes.txt
for sd in 1.5 1.7 1.9; do echo side $sd; MG_SIDE=$sd MG_SIZES=4096 python tools/mg_sizes.py; done
python tools/mgc_trace.py 256 2>&1 | grep "mgc trace" | tail -1
