#!/bin/bash
# developer tool: PMC passes over the multigrid smoothing kernels of a 4096^2 solve
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
TAG=${TAG:-march}
B="python $R/tools/mg_prof.py 4096"
n=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_WAVES SQ_BUSY_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${TAG}_g$n -- $B > $O/${TAG}_g$n.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for g in sorted(glob.glob("$O/${TAG}_g*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(g)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if "smooth" not in k: continue
        key = k[:60] + " grid=" + r.get("Grid_Size", "?")
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][r["Counter_Name"]] += 1
out = {k: {c: round(v / cnt[k][c]) for c, v in d.items()} | {"launches": max(cnt[k].values())} for k, d in acc.items()}
json.dump(out, open("$O/${TAG}_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*.csv" -size +3M -delete 2>/dev/null
