#!/bin/bash
# rocprofv3 kernel statistics + PMC passes (counters in passes of their own) over ONE kernel of a
# secondary leg -> gpurun_out/<TAG>_kernel_stats.csv, gpurun_out/<TAG>_pmc.json
#   LEG=swe KN=k_sw_wave NX=4096 CALLS_PER_STEP=1 TAG=r05_swe4096 bash tools/pmc_leg.sh
# CALLS_PER_STEP: launches of the kernel per time step (RK4: 4) -- per-cell figures are per LAUNCH.
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
LEG=${LEG:-swe}; KN=${KN:-k_sw_wave}; NX=${NX:-4096}; TAG=${TAG:-pmc_leg}; FM=${FM:-1}
B="python $R/tools/also_run.py $LEG"
export NX FM STEPS=${STEPS:-10}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- $B > $O/${TAG}_stats.log 2>&1
find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/${TAG}_kernel_stats.csv
n=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${TAG}_g$n -- $B > $O/${TAG}_g$n.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections, json, os
out = {}
for g in sorted(glob.glob("$O/${TAG}_g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(g)):
        if "$KN" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    out.update({k: round(v / cnt[k]) for k, v in acc.items()})
cells = float($NX) * float($NX)
if out.get("SQ_WAVES"):
    w = out["SQ_WAVES"]
    out["per_wave"] = {k: round(v / w, 1) for k, v in out.items() if k.startswith("SQ_INSTS") or k.startswith("SQ_ACTIVE") or k.startswith("SQ_WAIT") or k == "SQ_WAVE_CYCLES"}
    out["valu_lane_insts_per_cell_and_launch"] = out.get("SQ_INSTS_VALU", 0) * 64.0 / cells
# fabric bytes per launch, corrected as tools/make_traffic.py does (MI355X_MICROARCH.md, HBM section:
# FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of wide coalesced reads -> doubled)
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    rd, wr = out["FETCH_SIZE"] * 1024 * 2, out["WRITE_SIZE"] * 1024
    out["fabric_read_bytes_per_launch"], out["fabric_write_bytes_per_launch"] = rd, wr
    out["fabric_bytes_per_cell_and_launch"] = (rd + wr) / cells
if out.get("SQ_ACTIVE_INST_VALU") and out.get("GRBM_GUI_ACTIVE"):
    out["valu_busy_ms"] = out["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * 2.4e9) * 1e3
    out["kernel_ms_by_grbm"] = out["GRBM_GUI_ACTIVE"] / 8 / 2.4e9 * 1e3
    out["valu_busy_frac"] = out["valu_busy_ms"] / out["kernel_ms_by_grbm"]
# kernel time from the statistics pass
try:
    for r in csv.DictReader(open("$O/${TAG}_kernel_stats.csv")):
        if "$KN" in r["Name"]:
            out["kernel_avg_us"] = float(r["AverageNs"]) / 1e3; out["kernel_calls"] = int(r["Calls"]); break
except Exception:
    pass
out["config"] = {"leg": "$LEG", "nx": $NX, "fast_math": $FM, "kernel": "$KN"}
if os.environ.get("PYRO_PROVENANCE") and os.path.exists(os.environ["PYRO_PROVENANCE"]):
    out["provenance"] = json.load(open(os.environ["PYRO_PROVENANCE"]))
print(json.dumps(out, indent=1))
json.dump(out, open("$O/${TAG}_pmc.json", "w"), indent=1)
PY
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*.csv" -size +3M -delete 2>/dev/null
