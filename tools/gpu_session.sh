#!/bin/bash
# scratch: one GPU-box session (edited per use)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=r04j
python tools/box_check.py || { echo "slow box: giving the minutes back"; exit 0; }
cd /tmp
for dyn in -1 1; do
  B="python $GRAFT_REPO_ROOT/bench.py --nx 8192 --steps 5 --warmup 2 --no-also --no-cpu-baseline --fast-math 1 --kernel-set 2 --wave-dynamic $dyn"
  n=0
  for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
    n=$((n+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/$O/${TAG}_d${dyn}_g$n -- $B > $GRAFT_REPO_ROOT/$O/${TAG}_d${dyn}_g$n.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, json
for dyn in ("-1", "1"):
    out = {}
    for g in sorted(glob.glob("$O/${TAG}_d%s_g*/**/*counter_collection.csv" % dyn, recursive=True)):
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(g)):
            if "k_ctu_wave" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        out.update({k: v / cnt[k] for k, v in acc.items()})
    w = out.get("SQ_WAVES", 1)
    print("dyn", dyn, {k: round(v) for k, v in out.items()})
    print("   per wave:", {k: round(v / w, 1) for k, v in out.items() if k.startswith("SQ_")})
    print("   valu/cell", out.get("SQ_INSTS_VALU", 0) * 64 / 8192 ** 2, "valu busy", out.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / (out.get("GRBM_GUI_ACTIVE", 1) / 8),
          "kernel us", out.get("GRBM_GUI_ACTIVE", 0) / 8 / 2.4e3, "wait_any frac", out.get("SQ_WAIT_ANY", 0) / max(out.get("SQ_WAVE_CYCLES", 1), 1))
PY
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*.csv" -size +2M -delete 2>/dev/null
