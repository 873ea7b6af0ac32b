#!/bin/bash
# One GPU-box session: parity tests, default bench, rocprofv3 stats + PMC passes
# of the default bench command (both builds), 2-rank launcher check.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag>
TAG=${1:-r01}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
timeout 600 python bench.py --fast-math 0 --no-also --no-cpu-baseline > $O/bench_default_exact.json 2>> $O/bench_default.err
cd /tmp
for fm in 1 0; do
  B="python $R/bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline --fast-math $fm"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stats_fm$fm -- $B > $O/rocprof_${TAG}_fm$fm.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_${TAG}_fetch_fm$fm -- $B >> $O/rocprof_${TAG}_fm$fm.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_${TAG}_write_fm$fm -- $B >> $O/rocprof_${TAG}_fm$fm.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/prof_${TAG}_sq_fm$fm -- $B >> $O/rocprof_${TAG}_fm$fm.log 2>&1
done
# multigrid (10 V-cycles at 4096^2) and advection kernels: kernel stats of the `also` legs
MG_KINDS=15 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_mg_stats -- python $R/tools/mg_ab.py > $O/rocprof_${TAG}_mg.log 2>&1
cd $R
# launcher / N>1 plumbing on a 1-GPU box: two ranks share GPU 0 (debug only)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29533 bench.py --gpus 2 --nx 4096 --steps 5 --warmup 2 --no-also --no-cpu-baseline \
  > $O/bench_2rank.json 2> $O/bench_2rank.err
tail -c 400 $O/bench_2rank.json; tail -5 $O/bench_2rank.err
# keep only the small csv files of the profiles
find $O -name "*.db" -delete 2>/dev/null
du -sh $O | tail -1
