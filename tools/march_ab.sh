#!/bin/bash
# A/B of the fused kernel sets on the GPU box (developer tool): parity tests of the
# compressible solver, then bench legs per kernel set / build / size
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=${TAG:-march}
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1200 python -m pytest tests/test_device_compressible.py -m gpu -x -q > $O/${TAG}_pytest.log 2>&1
  tail -3 $O/${TAG}_pytest.log
fi
for nx in ${SIZES:-16384 8192 4096}; do
 for ks in ${KSETS:-1 2}; do
  for fm in ${FMS:-1 0}; do
    timeout 300 python bench.py --nx $nx --kernel-set $ks --fast-math $fm --no-also --no-cpu-baseline --steps ${STEPS:-20} > $O/${TAG}_b_${nx}_k${ks}_f${fm}.json 2> $O/${TAG}_b_${nx}_k${ks}_f${fm}.err
    python - <<PY
import json
try:
    d = json.load(open("$O/${TAG}_b_${nx}_k${ks}_f${fm}.json"))
    k = d["roofline"]["kernels"]
    print("nx=$nx kset=$ks fast=$fm  ms/step=%.3f  Gcell/s=%.2f " % (d["ms_per_step"], d["value"]/1e9), {a: round(b["avg_ms"], 3) for a, b in k.items()})
except Exception as e:
    print("nx=$nx kset=$ks fast=$fm FAILED", e, open("$O/${TAG}_b_${nx}_k${ks}_f${fm}.err").read()[-500:])
PY
  done
 done
done
