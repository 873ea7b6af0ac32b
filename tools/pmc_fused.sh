#!/bin/bash
# PMC passes over the default compressible step at NX^2 (developer tool)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
NX=${NX:-8192}; FM=${FM:-1}; TAG=${TAG:-pmc}
B="python $R/bench.py --nx $NX --steps 5 --warmup 2 --no-also --no-cpu-baseline --fast-math $FM"
n=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"; do
  n=$((n+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${TAG}_g$n -- $B > $O/${TAG}_g$n.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
for g in sorted(glob.glob("$O/${TAG}_g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(g)):
        if "k_ctu_fused" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    print(g.split("/")[-3], {k: round(v / cnt[k]) for k, v in acc.items()})
PY
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*.csv" -size +3M -delete 2>/dev/null
