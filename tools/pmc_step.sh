#!/bin/bash
# PMC passes over the compressible step kernel (developer tool)
#   NX=8192 FM=1 KS=2 TAG=pmc bash tools/pmc_step.sh
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
NX=${NX:-8192}; FM=${FM:-1}; KS=${KS:-2}; TAG=${TAG:-pmc}
KN=$(case "$KS" in 1) echo k_ctu_fused;; *) echo k_ctu_wave;; esac)
B="python $R/bench.py --nx $NX --steps 5 --warmup 2 --no-also --no-cpu-baseline --fast-math $FM --kernel-set $KS"
n=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM" \
           ${EXTRA_GROUPS:+"$EXTRA_GROUPS"}; do
  n=$((n+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${TAG}_g$n -- $B > $O/${TAG}_g$n.log 2>&1
done
if [ "${TRAFFIC:-0}" = "1" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    n=$((n+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${TAG}_g$n -- $B > $O/${TAG}_g$n.log 2>&1
  done
fi
cd $R
python - <<PY
import csv, glob, collections, json
out = {}
for g in sorted(glob.glob("$O/${TAG}_g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(g)):
        if "$KN" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    out.update({k: round(v / cnt[k]) for k, v in acc.items()})
if out.get("SQ_WAVES"):
    w = out["SQ_WAVES"]
    out["per_wave"] = {k: round(v / w, 1) for k, v in out.items() if k.startswith("SQ_INSTS") or k.startswith("SQ_ACTIVE") or k.startswith("SQ_WAIT") or k == "SQ_WAVE_CYCLES"}
    out["valu_per_cell_update"] = out.get("SQ_INSTS_VALU", 0) * 64.0 / ($NX * $NX)
out["config"] = {"nx": $NX, "fast_math": $FM, "kernel_set": $KS, "kernel": "$KN"}
print(json.dumps(out, indent=1))
json.dump(out, open("$O/${TAG}_summary.json", "w"), indent=1)
PY
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*.csv" -size +3M -delete 2>/dev/null
