#!/bin/bash
# PMC passes + kernel trace of the advection step kernel (developer tool)
#   NX=2048 ROWS=0 bash tools/pmc_adv.sh            (one step per launch: k_adv_step)
#   NX=2048 ROWS=0 MULTI=2 bash tools/pmc_adv.sh    (MULTI steps per launch: k_adv_multi)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
NX=${NX:-2048}; ROWS=${ROWS:-0}; TAG=${TAG:-pmcadv}
MULTI=${MULTI:-0}; KPAT=k_adv_step
B="python $R/tools/adv_time.py"
export SIZES="$NX:$ROWS"
if [ "$MULTI" != "0" ]; then B="python $R/tools/adv_multi_time.py"; export SPEC="$NX:$MULTI/$ROWS"; KPAT=k_adv_multi; fi
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- $B > $O/${TAG}_stats.log 2>&1
n=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${TAG}_g$n -- $B > $O/${TAG}_g$n.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections, json
out = {}
for g in sorted(glob.glob("$O/${TAG}_g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(g)):
        if "$KPAT" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    out.update({k: v / cnt[k] for k, v in acc.items()})
for g in glob.glob("$O/${TAG}_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        if "$KPAT" in r["Name"]:
            out["kernel_trace_avg_us"] = float(r["AverageNs"]) / 1e3
            out["kernel_trace_calls"] = int(r["Calls"])
nx = $NX
spl = max($MULTI, 1)          # time steps per launch
out["steps_per_launch"] = spl
if out.get("SQ_WAVES"):
    w = out["SQ_WAVES"]
    out["valu_per_cell"] = out["SQ_INSTS_VALU"] * 64 / (nx * nx) / spl
    out["valu_busy"] = out["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (out["GRBM_GUI_ACTIVE"] / 8)
    out["kernel_us_from_grbm"] = out["GRBM_GUI_ACTIVE"] / 8 / 2.4e3
    out["wave_cycles_per_wave"] = out["SQ_WAVE_CYCLES"] * 4 / w
    out["wait_any_frac"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
    out["wait_inst_frac"] = out["SQ_WAIT_INST_ANY"] / out["SQ_WAVE_CYCLES"]
    out["read_bytes"] = out.get("FETCH_SIZE", 0) * 1024 * 2
    out["write_bytes"] = out.get("WRITE_SIZE", 0) * 1024
    out["traffic_over_algorithmic"] = (out["read_bytes"] + out["write_bytes"]) / (16.0 * nx * nx * spl)
    out["traffic_bytes_per_cell_per_step"] = (out["read_bytes"] + out["write_bytes"]) / (nx * nx * spl)
print(json.dumps(out, indent=1))
json.dump(out, open("$O/${TAG}_summary.json", "w"), indent=1)
PY
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*.csv" -size +3M -delete 2>/dev/null
