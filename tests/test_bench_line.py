"""The ONE stdout line of bench.py (the driver's record of a round) stays parsable.

Round 4's line carried every secondary leg and grew to ~30 KB: the driver's bounded tail cut
it and `BENCH_r04.parsed` was null.  The line is now compact (< 4 KB: headline, roofline,
cpu_baseline, one figure per secondary leg); the full record goes to a side file / stderr."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def check_line(line):
    assert len(line) < 4096, len(line)
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    assert d["config"]["workload"].startswith("compressible sedov")
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_ms"):
        assert k in rf, k
    assert rf["frac"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 * rf["frac"] + 1e-12
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    return d


def test_bench_whole_script_on_the_emulator(tmp_path):
    """the driver's exact flags (+ a tiny grid and every secondary leg scaled down) through
    bench.main() on the host emulator: ONE stdout line, < 4 KB, with roofline, cpu_baseline and
    targets; the full record lands in the side file"""
    env = dict(os.environ, PYRO_TEST_WORKERS="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_bench_emu.py"),
                        "--gpus", "1", "--steps", "20", "--warmup", "5", "--nx", "64", "--also-div", "64",
                        "--cpu-sample-nx", "32", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = check_line(lines[0])
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1
    t = d["targets"]
    for k in ("sedov_exact", "sedov_developed", "sedov_4096", "sedov_8192", "advection_2048", "advection_8192",
              "mg_4096", "incompressible_2048", "pyro_driver"):
        assert k in t and "error" not in t[k], (k, t.get(k))
    for k, v in t["pyro_driver"].items():
        assert "error" not in v, (k, v)
    full = json.load(open(os.path.join(ROOT, d["full_record"])))
    assert "also" in full and abs(full["value"] - d["value"]) <= 1e-5 * d["value"]
    assert "[bench also] multigrid:" in p.stderr


@pytest.mark.parametrize("name", ["r04_bench_default.json"])
def test_compact_line_of_a_full_hardware_record(name):
    """a full record as a GPU box produced it (round 4's 30 KB line) through compact_line"""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", name)))
    d = check_line(bench.compact_line(full, "gpurun_out/bench_full.json"))
    assert d["roofline"]["kernel"] == "k_ctu_wave" and d["roofline"]["traffic"] > 0
    assert d["targets"]["sedov_8192"]["frac"] > 0.1 and d["targets"]["advection_2048"]["step_frac"] > 0.3
    assert d["targets"]["mg_4096"]["value"] > 1000


def test_compact_line_with_eight_ranks():
    """the N = 8 line carries the per-rank table and still fits"""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    full.pop("also")
    cols = ("kernel_ms", "halo_wait_ms", "halo_sync_ms", "allreduce_ms", "stream_ms_per_step", "rows",
            "col_strips", "rows_per_strip", "row_strips", "overlap", "wavefronts")
    table = {c: [1.2345678901 + i for i in range(8)] for c in cols}
    full["n_gpus"] = 8
    full["ranks"] = {"per_rank": table, "min": {c: 1.0 for c in cols}, "max": {c: 9.0 for c in cols},
                     "note": "x" * 600, "predicted_ms_per_step": 1.4, "predicted_source": "DESIGN"}
    full["config"]["scale_check"] = {"ok": True, "nx": 2048, "steps": 12, "note": "y" * 500}
    d = check_line(bench.compact_line(full, None))
    assert len(d["ranks"]["kernel_ms"]) == 8
