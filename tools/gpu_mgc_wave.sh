#!/bin/bash
# developer session: the coarse kernel's wave-resident levels (tests, trace, A/B timing)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_device_multigrid.py -m gpu -x -q -k "coarse_wave or march_tails or poisson or regression" > gpurun_out/mgw_tests.log 2>&1; tail -3 gpurun_out/mgw_tests.log
python tools/mgc_trace.py 256 2>&1 | grep "mgc trace" | tail -2
for w in 1 0; do MG_WAVE=$w MG_SIZES=64,128,512,2048,4096 timeout 300 python tools/mg_sizes.py > gpurun_out/mgw_sizes_wave$w.txt 2>&1; cat gpurun_out/mgw_sizes_wave$w.txt; done
