#!/bin/bash
# One GPU-box session of round 3 (developer tool).  Steps by environment:
#   TESTS=1    pytest -m gpu (all, or -k "$TESTK")
#   BENCH=1    default bench -> gpurun_out/bench_${TAG}.json
#   PMC="1 0"  PMC passes of the 16384^2 step kernel for these builds (TRAFFIC=1 adds FETCH/WRITE)
#   STATS="1"  rocprofv3 --kernel-trace --stats of the default bench command for these builds
#   ALSOSTATS=1  rocprofv3 kernel stats of the advection / multigrid legs
# usage (repo root on the GPU box): TAG=r03a TESTS=1 BENCH=1 PMC="1" bash tools/gpu_r03.sh
TAG=${TAG:-r03}
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
if [ "${TESTS:-0}" = "1" ]; then
  ( time timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q ${TESTK:+-k "$TESTK"} ) > $O/pytest_gpu_${TAG}.log 2>&1
  tail -6 $O/pytest_gpu_${TAG}.log
fi
if [ "${BENCH:-0}" = "1" ]; then
  ( time timeout 1200 python bench.py ${BENCH_ARGS} > $O/bench_${TAG}.json ) 2> $O/bench_${TAG}.err
  head -c 600 $O/bench_${TAG}.json; echo; tail -4 $O/bench_${TAG}.err
fi
for fm in ${PMC}; do
  TRAFFIC=${TRAFFIC:-0} NX=${PMC_NX:-16384} FM=$fm KS=-1 TAG=pmc_${TAG}_fm$fm bash tools/pmc_step.sh > $O/pmc_${TAG}_fm$fm.txt 2>&1
  tail -25 $O/pmc_${TAG}_fm$fm.txt
done
cd /tmp
for fm in ${STATS}; do
  B="python $R/bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline --fast-math $fm"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stats_fm$fm -- $B > $O/rocprof_${TAG}_fm$fm.log 2>&1
  find $O/prof_${TAG}_stats_fm$fm -name "*kernel_stats.csv" | head -1 | xargs -r head -8
done
if [ "${ALSOSTATS:-0}" = "1" ]; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_adv -- python $R/tools/also_run.py adv > $O/rocprof_${TAG}_adv.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_mg -- python $R/tools/also_run.py mg > $O/rocprof_${TAG}_mg.log 2>&1
  for d in adv mg; do find $O/prof_${TAG}_$d -name "*kernel_stats.csv" | head -1 | xargs -r head -12; done
fi
cd $R
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
du -sh $O | tail -1
