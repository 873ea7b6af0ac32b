cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
TAG=r03k TESTS=1 BENCH=1 PMC="1 0" TRAFFIC=1 STATS="1 0" ALSOSTATS=1 bash tools/gpu_r03.sh > $O/r03k_session.log 2>&1
tail -5 $O/pytest_gpu_r03k.log
TAG=r03k bash tools/pmc_also.sh > $O/r03k_pmc_also.log 2>&1
MG_SIZES=512,1024,2048,4096 timeout 300 python tools/mg_sizes.py > $O/r03k_mg_sizes.txt 2>&1; cat $O/r03k_mg_sizes.txt
timeout 600 python bench.py --gpus 2 --nx 4096 --steps 5 --warmup 2 --no-also --no-cpu-baseline --scale-check > $O/r03k_bench_2rank.json 2> $O/r03k_bench_2rank.err; tail -c 600 $O/r03k_bench_2rank.json; tail -3 $O/r03k_bench_2rank.err
head -c 400 $O/bench_r03k.json
