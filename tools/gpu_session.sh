#!/bin/bash
# scratch: one GPU-box session (edited per use)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
TAG=r04f
( echo "--- product (one strip per workgroup)"; MG_SIZES=2048,4096 python tools/mg_sizes.py
  for g in g2 g4; do for d in 0 2 4 8; do
    echo "--- $g sync $d"; PYRO2_AMD_LIB=$PWD/pyro2_amd/lib/libpyrohip_$g.so MG_SYNC=$d MG_SIZES=2048,4096 python tools/mg_sizes.py
  done; done
) > $O/${TAG}_mg_group.txt 2>&1
cat $O/${TAG}_mg_group.txt
PYRO2_AMD_LIB=$PWD/pyro2_amd/lib/libpyrohip_g2.so TAG=march_${TAG}_g2 bash tools/pmc_march.sh > $O/march_${TAG}_g2.txt 2>&1
python - <<PY
import json
for g in ("g2",):
    d = json.load(open("$O/march_${TAG}_%s_pmc.json" % g))
    for k, v in d.items():
        if "march" in k:
            print(g, k[-40:], "launches", v["launches"], "read MB", v.get("FETCH_SIZE", 0) * 2 / 1024, "write MB", v.get("WRITE_SIZE", 0) / 1024,
                  "valu busy", v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / max(v.get("GRBM_GUI_ACTIVE", 1) / 8, 1), "us", v.get("GRBM_GUI_ACTIVE", 0) / 8 / 2.4e3)
PY
