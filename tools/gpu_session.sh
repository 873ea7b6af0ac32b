cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
PYRO_BENCH_COMM=host timeout 600 python bench.py --gpus 2 --nx 4096 --steps 5 --warmup 2 --no-also --no-cpu-baseline --scale-check > $O/r03z_bench_2rank.json 2> $O/r03z_bench_2rank.err; tail -c 900 $O/r03z_bench_2rank.json; tail -3 $O/r03z_bench_2rank.err
