#!/bin/bash
# fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and LDS / VALU activity of the
# advection step (2048^2) and the multigrid V-cycle (4096^2) -> profiles/<tag>_also_traffic.json
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
TAG=${TAG:-r02}
for what in adv mg; do
  n=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    n=$((n+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/also_${what}_g$n -- python $R/tools/also_run.py $what > $O/also_${what}_g$n.log 2>&1
  done
done
cd $R
python - <<PY
import csv, glob, collections, json
out = {}
for what, unit_launches in (("adv", None), ("mg", None)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for g in sorted(glob.glob(f"$O/also_{what}_g*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(g)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            per[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    ks = {}
    for k, d in per.items():
        if not (k.startswith("pyro::k_mg") or "k_adv" in k or "k_vc" in k):
            continue
        n = max(cnt[k].values())
        e = {"launches": n}
        for c, v in d.items():
            e[c + "_total"] = v
        e["read_bytes_total"] = d.get("FETCH_SIZE", 0) * 1024 * 2     # gfx950: half of coalesced reads reported
        e["write_bytes_total"] = d.get("WRITE_SIZE", 0) * 1024
        ks[k] = e
    out[what] = ks
import os
if os.environ.get("PYRO_PROVENANCE"):
    out["provenance"] = json.load(open(os.environ["PYRO_PROVENANCE"]))
# advection (also_run.py adv = the bench leg at 2048^2): the several-steps-per-launch kernel
# k_adv_multi takes STEPS_PER_LAUNCH time steps per launch
import re
multi = sorted((k for k in out["adv"] if "k_adv_multi" in k), key=lambda k: -out["adv"][k]["launches"])
if multi:
    k = multi[0]                                  # the instance the leg launches most
    spl = int(re.findall(r"(\d+)>", k)[-1])       # its last template argument: steps per launch
    e = out["adv"][k]
    per_launch = (e["read_bytes_total"] + e["write_bytes_total"]) / e["launches"]
    out["adv_summary"] = {"kernel": k, "nx": 2048, "steps_per_launch": spl,
                          "bytes_per_launch_2048": per_launch, "bytes_per_step": per_launch / spl,
                          "algorithmic_bytes_per_step": 16 * 2048 * 2048,
                          "valu_insts_per_cell_update": e.get("SQ_INSTS_VALU_total", 0) * 64 / e["launches"] / (2048 * 2048) / spl,
                          "valu_busy_frac": e.get("SQ_ACTIVE_INST_VALU_total", 0) * 4 / 1024 / max(e.get("GRBM_GUI_ACTIVE_total", 1) / 8, 1)}
# multigrid: all kernels of the 12 V-cycles the leg runs (2 warm-up + 10 timed)
if out["mg"]:
    tot_r = sum(e["read_bytes_total"] for e in out["mg"].values())
    tot_w = sum(e["write_bytes_total"] for e in out["mg"].values())
    lds = sum(e.get("SQ_ACTIVE_INST_LDS_total", 0) for e in out["mg"].values())
    valu = sum(e.get("SQ_ACTIVE_INST_VALU_total", 0) for e in out["mg"].values())
    busy = sum(e.get("GRBM_GUI_ACTIVE_total", 0) for e in out["mg"].values()) / 8
    out["mg_summary"] = {"nx": 4096, "vcycles_profiled": 12, "bytes_per_vcycle": (tot_r + tot_w) / 12,
                         "read_bytes_per_vcycle": tot_r / 12, "write_bytes_per_vcycle": tot_w / 12,
                         "model_bytes_per_vcycle": 720 * 4096 * 4096,
                         "bytes_per_finest_cell_per_vcycle": (tot_r + tot_w) / 12 / (4096 * 4096),
                         "lds_active_frac_of_gpu_cycles": lds * 4 / 1024 / max(busy, 1),
                         "valu_active_frac_of_gpu_cycles": valu * 4 / 1024 / max(busy, 1),
                         "note": "fabric bytes (FETCH_SIZE doubled, WRITE_SIZE as reported) of every multigrid kernel "
                                 "of the bench leg (setup + solve: 2 warm-up + 10 timed V-cycles); LDS / VALU: "
                                 "SQ_ACTIVE_INST_* quad-cycles summed over the SIMDs / (1024 SIMDs x busy cycles)"}
json.dump(out, open("$O/${TAG}_also_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.endswith("summary")}, indent=1))
PY
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*.csv" -size +3M -delete 2>/dev/null
