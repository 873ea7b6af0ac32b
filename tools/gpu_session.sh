#!/bin/bash
# scratch: one GPU-box session (edited per use)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
TAG=r04a
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_${TAG}.log 2>&1; tail -4 $O/pytest_gpu_${TAG}.log
# advection: steps per launch
( SPEC="2048:0/0,1/0,2/0,2/13,2/26,2/38,3/0;4096:0/0,2/0,3/0;8192:0/0,2/0,2/141,2/95,3/0" CHECK=1 python tools/adv_multi_time.py
  SPEC="2048:2/0,3/0;8192:2/0,3/0" PRIO=1 python tools/adv_multi_time.py
  echo "--- exact build"; SPEC="2048:0/0,2/0,3/0;8192:0/0,2/0" FAST=0 CHECK=1 python tools/adv_multi_time.py
  echo "--- K=3 one wavefront per SIMD"; PYRO2_AMD_LIB=$PWD/pyro2_amd/lib/libpyrohip_w1.so SPEC="2048:3/0,3/38;8192:3/0" python tools/adv_multi_time.py
  echo "--- non-temporal stores"; PYRO2_AMD_LIB=$PWD/pyro2_amd/lib/libpyrohip_nt.so SPEC="2048:2/0,3/0;8192:2/0" python tools/adv_multi_time.py
) > $O/${TAG}_adv_multi.txt 2>&1
cat $O/${TAG}_adv_multi.txt
# Sedov kernel statistics + counters at the north_star size and config 3's
for nx in 8192 4096; do
  TRAFFIC=1 NX=$nx FM=1 KS=-1 TAG=pmc_${TAG}_sedov$nx bash tools/pmc_step.sh > $O/pmc_${TAG}_sedov$nx.txt 2>&1
  tail -30 $O/pmc_${TAG}_sedov$nx.txt | head -40
  ( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_${TAG}_sedov$nx -- python $GRAFT_REPO_ROOT/bench.py --nx $nx --steps 10 --warmup 3 --no-also --no-cpu-baseline --fast-math 1 > $GRAFT_REPO_ROOT/$O/rocprof_${TAG}_sedov$nx.log 2>&1 )
  find $O/prof_${TAG}_sedov$nx -name "*kernel_stats.csv" | head -1 | xargs -r head -8
done
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
du -sh $O | tail -1
