#!/bin/bash
# scratch: one GPU-box session (edited per use)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
TAG=r04d
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_${TAG}.log 2>&1; grep -E "passed|failed|rror" $O/pytest_gpu_${TAG}.log | tail -3
( time timeout 1500 python bench.py > $O/bench_${TAG}.json ) 2> $O/bench_${TAG}.err; tail -4 $O/bench_${TAG}.err
python - <<PY
import json
d = json.load(open("$O/bench_${TAG}.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("cpu_baseline", {}).get("reference_numpy_stages_only"))
a = d["also"]
for k in ("advection", "advection_8192"):
    print(k, a[k]["ms_per_step"], a[k]["steps_per_launch"], a[k]["roofline"]["frac"], a[k]["roofline"]["step_frac"], a[k]["ms_per_step_one_launch_per_step"], a[k]["other_build"]["roofline"]["step_frac"])
print(json.dumps(a.get("pyro_driver"), indent=1)[:3000])
PY
PYRO_BENCH_COMM=host timeout 600 python bench.py --gpus 2 --nx 4096 --steps 5 --warmup 2 --no-also --no-cpu-baseline > $O/bench_${TAG}_2rank.json 2> $O/bench_${TAG}_2rank.err
tail -c 1800 $O/bench_${TAG}_2rank.json; tail -5 $O/bench_${TAG}_2rank.err
