#!/usr/bin/env python3
"""Summarise rocprofv3 outputs under gpurun_out/ into small tracked files
under profiles/ (kernel stats CSV + per-kernel PMC averages as JSON).

usage: tools/summarize_rocprof.py <tag> [--stats DIR] [--pmc DIR ...]
FETCH_SIZE is doubled (gfx950 reports half of the bytes of wide coalesced
reads, MI355X_MICROARCH.md "HBM"; calibrated here on k_prim: 4 planes read,
FETCH_SIZE = 2 planes)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main():
    tag = sys.argv[1]
    args = sys.argv[2:]
    out = {}
    i = 0
    while i < len(args):
        if args[i] == "--stats":
            for f in glob.glob(os.path.join(args[i + 1], "**", "*kernel_stats.csv"), recursive=True):
                shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
            i += 2
        elif args[i] == "--pmc":
            for f in glob.glob(os.path.join(args[i + 1], "**", "*counter_collection.csv"), recursive=True):
                agg = collections.defaultdict(lambda: collections.defaultdict(float))
                cnt = collections.defaultdict(lambda: collections.Counter())
                meta = {}
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"].split("(")[0]
                    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    cnt[k][r["Counter_Name"]] += 1
                    meta[k] = {"vgpr": int(r["VGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                               "wg": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"])}
                for k in agg:
                    d = out.setdefault(k, dict(meta[k]))
                    for c, v in agg[k].items():
                        d[c + "_per_launch"] = v / cnt[k][c]
                        d["launches"] = cnt[k][c]
            i += 2
        else:
            i += 1
    for k, d in out.items():
        if "FETCH_SIZE_per_launch" in d:
            d["hbm_read_bytes_per_launch"] = d["FETCH_SIZE_per_launch"] * 1024 * 2
        if "WRITE_SIZE_per_launch" in d:
            d["hbm_write_bytes_per_launch"] = d["WRITE_SIZE_per_launch"] * 1024
    if out:
        json.dump(out, open(f"profiles/{tag}_pmc.json", "w"), indent=1, sort_keys=True)
    print("wrote profiles/" + tag + "_*")


if __name__ == "__main__":
    main()
