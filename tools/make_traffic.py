#!/usr/bin/env python3
"""profiles/traffic.json from the per-kernel PMC summaries written by
tools/summarize_rocprof.py.  usage: make_traffic.py <nx> <pmc_fm1.json> <pmc_fm0.json> <stats_fm1.csv> <stats_fm0.csv>"""
import csv
import json
import sys


def kernel_ms(stats_csv, name):
    for r in csv.DictReader(open(stats_csv)):
        if name in r["Name"]:
            return float(r["AverageNs"]) * 1e-6
    return None


def main():
    nx = int(sys.argv[1])
    out = {}
    for key, pmc, stats in (("fast_math_1", sys.argv[2], sys.argv[4]),
                            ("fast_math_0", sys.argv[3], sys.argv[5])):
        d = json.load(open(pmc))
        k = next(n for n in d if "k_ctu_fused" in n)
        e = d[k]
        rd, wr = e["hbm_read_bytes_per_launch"], e["hbm_write_bytes_per_launch"]
        ms = kernel_ms(stats, "k_ctu_fused")
        # fp64 VALU: one wave instruction occupies a SIMD for 4 cycles (16 lanes);
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs
        valu_ms = e["SQ_ACTIVE_INST_VALU_per_launch"] * 4 / (1024 * 2.4e9) * 1e3
        out[key] = {
            "kernel": "k_ctu_fused",
            "measured_at": f"sedov {nx}x{nx}, 1 MI355X, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                           "(separate passes)",
            "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
            "bytes_per_cell_update": (rd + wr) / (nx * nx),
            "note": "FETCH_SIZE doubled (gfx950 reports half of coalesced read bytes; calibrated "
                    "on k_prim: 4 planes read, 2 reported), WRITE_SIZE as reported; "
                    "Infinity-Cache hits are counted, so this is fabric traffic >= HBM traffic",
            "valu_insts_per_wave": e["SQ_INSTS_VALU_per_launch"] / e["SQ_WAVES_per_launch"],
            "valu_busy_ms": valu_ms, "kernel_ms": ms,
        }
    json.dump(out, open("profiles/traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
