#!/bin/bash
# One GPU-box session: parity tests, default bench, rocprofv3 kernel stats + PMC passes
# of the default bench command (both builds), multi-rank launcher check.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag>
TAG=${1:-r02}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
  tail -3 $O/pytest_gpu.log
fi
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 800 $O/bench_default.json; tail -3 $O/bench_default.err
cd /tmp
for fm in 1 0; do
  B="python $R/bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline --fast-math $fm"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stats_fm$fm -- $B > $O/rocprof_${TAG}_fm$fm.log 2>&1
done
cd $R
for fm in 1 0; do
  TRAFFIC=1 NX=16384 FM=$fm KS=-1 TAG=pmc_${TAG}_fm$fm bash tools/pmc_step.sh > $O/pmc_${TAG}_fm$fm.txt 2>&1
done
# launcher / N>1 plumbing on a 1-GPU box: two ranks share GPU 0 (debug only)
timeout 600 python bench.py --gpus 2 --nx 4096 --steps 5 --warmup 2 --no-also --no-cpu-baseline \
  > $O/bench_2rank.json 2> $O/bench_2rank.err
tail -c 400 $O/bench_2rank.json; tail -5 $O/bench_2rank.err
find $O -name "*.db" -delete 2>/dev/null
du -sh $O | tail -1
