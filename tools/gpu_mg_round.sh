#!/bin/bash
# GPU-box session after a change to the multigrid kernels: parity tests, smoke(), default
# bench, V-cycle time by size, rocprofv3 kernel statistics and PMC passes of the multigrid
# and advection legs.  usage: bash tools/gpu_mg_round.sh <tag>
TAG=${1:-r02h}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 600 $O/bench_default.json
python tools/mg_prof.py 512 1024 2048 4096 > $O/${TAG}_mg_vcycle_by_size.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_mg -- python $R/tools/also_run.py mg > $O/rocprof_${TAG}_mg.log 2>&1
cd $R
TAG=$TAG bash tools/pmc_also.sh > $O/${TAG}_pmc_also.txt 2>&1
TAG=${TAG}_march bash tools/pmc_march.sh > $O/${TAG}_pmc_march.txt 2>&1
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
du -sh $O | tail -1
