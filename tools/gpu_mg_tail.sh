#!/bin/bash
# developer session: the marching smoother's tails (tests, A/B timing, kernel statistics)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
[ -n "$SKIPTESTS" ] || { timeout 900 python -m pytest tests/test_device_multigrid.py -m gpu -x -q > gpurun_out/mgt_tests.log 2>&1; tail -3 gpurun_out/mgt_tests.log; }
for t in 1 0; do MG_TAIL=$t MG_SIZES=2048,4096 timeout 300 python tools/mg_sizes.py > gpurun_out/mgt_sizes_tail$t.txt 2>&1; cat gpurun_out/mgt_sizes_tail$t.txt; done
R=$PWD
(cd /tmp && MG_SIZES=4096 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mgt_prof -o mgt -- python $R/tools/mg_sizes.py > $R/gpurun_out/mgt_prof.log 2>&1)
find gpurun_out/mgt_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/mgt_kernel_stats.csv
head -12 gpurun_out/mgt_kernel_stats.csv | cut -c1-160
