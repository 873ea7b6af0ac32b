#!/bin/bash
# scratch: one GPU-box session (edited per use)
cd "$GRAFT_REPO_ROOT"
TAG=r03o TESTS=1 BENCH=1 ALSOSTATS=1 bash tools/gpu_r03.sh
TAG=r03o bash tools/pmc_also.sh > gpurun_out/pmc_also_r03o.log 2>&1; grep -A12 '"mg_summary"' gpurun_out/pmc_also_r03o.log | head -8
python tools/mg_sizes.py > gpurun_out/r03o_mg_sizes.txt 2>&1; cat gpurun_out/r03o_mg_sizes.txt
