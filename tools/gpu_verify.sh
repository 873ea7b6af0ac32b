#!/bin/bash
# Short GPU-box session: parity tests, smoke(), default bench, kernel-trace stats
# of the default bench command.  usage: bash tools/gpu_verify.sh <tag>
TAG=${1:-r01}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 900 $O/bench_default.json
cd /tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stats -- $B > $O/rocprof_${TAG}.log 2>&1
cd $R
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
du -sh $O | tail -1
