#!/bin/bash
# scratch: one GPU-box session (edited per use)
cd "$GRAFT_REPO_ROOT"
TAG=r03u TESTS=1 BENCH=1 ALSOSTATS=1 bash tools/gpu_r03.sh
TAG=r03u bash tools/pmc_also.sh > gpurun_out/pmc_also_r03x.log 2>&1; tail -12 gpurun_out/pmc_also_r03x.log | head -10
python tools/mg_sizes.py > gpurun_out/r03x_mg_sizes.txt 2>&1; cat gpurun_out/r03x_mg_sizes.txt
