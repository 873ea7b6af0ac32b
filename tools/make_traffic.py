#!/usr/bin/env python3
"""traffic.json (in the current directory) from tools/pmc_step.sh summaries (TRAFFIC=1) of the
compressible step: usage: make_traffic.py <summary.json> ...   One entry per summary, keyed
fast_math_<fm> for the 16384^2 headline and nx<N>_fast_math_<fm> for the other sizes, each with
the provenance (commit, box, date) of the session that counted it (PYRO_PROVENANCE = path of
the session's provenance file, tools/profile_round.sh)."""
import os
import json
import sys


def entry(d):
    nx = d["config"]["nx"]
    cells = float(nx) * nx
    rd = d.get("FETCH_SIZE", 0) * 1024 * 2       # gfx950: FETCH_SIZE reports half of coalesced reads
    wr = d.get("WRITE_SIZE", 0) * 1024
    # one FP64 wave instruction occupies a SIMD for 4 cycles (16 lanes);
    # SQ_ACTIVE_INST_VALU counts those quad-cycles summed over the SIMDs
    valu_ms = d["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * 2.4e9) * 1e3
    kernel_ms = d["GRBM_GUI_ACTIVE"] / 8 / 2.4e9 * 1e3   # 8 XCDs count, 2.4 GHz
    flops = (d.get("SQ_INSTS_VALU_ADD_F64", 0) + d.get("SQ_INSTS_VALU_MUL_F64", 0) +
             2 * d.get("SQ_INSTS_VALU_FMA_F64", 0) + d.get("SQ_INSTS_VALU_TRANS_F64", 0)) * 64
    return {
        "kernel": d["config"]["kernel"],
        "measured_at": f"sedov {nx}x{nx} (bench.py default state), 1 MI355X, rocprofv3 --pmc, "
                       "separate passes per counter group (tools/pmc_step.sh)",
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
        "bytes_per_cell_update": (rd + wr) / cells,
        "note": "FETCH_SIZE doubled (gfx950 reports half of coalesced read bytes; calibrated in "
                "round 1 on k_prim: 4 planes read, 2 reported), WRITE_SIZE as reported; "
                "Infinity-Cache hits are counted, so this is fabric traffic >= HBM traffic",
        "waves": d["SQ_WAVES"],
        "valu_insts_per_wave": d["SQ_INSTS_VALU"] / d["SQ_WAVES"],
        "valu_insts_per_cell_update": d["SQ_INSTS_VALU"] * 64 / cells,
        "flops_per_cell_update": flops / cells,
        "trans_f64_per_cell_update": d.get("SQ_INSTS_VALU_TRANS_F64", 0) * 64 / cells,
        "valu_busy_ms": valu_ms, "kernel_ms": kernel_ms,
    }


def main():
    prov = None
    if os.environ.get("PYRO_PROVENANCE"):
        prov = json.load(open(os.environ["PYRO_PROVENANCE"]))
    out = {"provenance": prov}
    for path in sys.argv[1:]:
        d = json.load(open(path))
        nx, fm = d["config"]["nx"], d["config"]["fast_math"]
        key = f"fast_math_{fm}" if nx == 16384 else f"nx{nx}_fast_math_{fm}"
        out[key] = dict(entry(d), nx=nx, provenance=prov)
    json.dump(out, open("traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
