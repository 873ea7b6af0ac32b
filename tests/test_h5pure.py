"""util/h5pure.py: pure Python HDF5 subset ("h5py-free
path").  Pinned three ways:
  * reads a file written by the real h5py / libhdf5 (tests/golden/h5py_written.h5,
    generator: oracle/gen_h5_fixture.py) and reproduces its content exactly;
  * write -> read round trip of pyro's output layout and the corner cases;
  * where an interpreter with h5py exists (this container: /opt/conda/bin/python3.9)
    a file written by h5pure is read back with the real h5py, and pyro's own
    shipped benchmark files are read by h5pure (where /root/reference exists).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pyro2_amd.util import h5pure

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
CONDA = "/opt/conda/bin/python3.9"
REF = "/root/reference/pyro"


def _eq(a, b):
    if isinstance(b, str):
        return isinstance(a, str) and a == b
    return np.array_equal(np.asarray(a), np.asarray(b))


def test_reads_file_written_by_h5py():
    exp = json.load(open(os.path.join(GOLD, "h5py_written.json")))
    with h5pure.File(os.path.join(GOLD, "h5py_written.h5")) as f:
        assert list(f) == ["BC", "aux", "empty", "grid", "misc", "runtime parameters", "state"]
        assert len(f["state"]) == 18 and len(f["empty"]) == 0
        for path, attrs in exp["attrs"].items():
            got = f.attrs if path == "/" else f[path].attrs
            assert sorted(got) == sorted(attrs)
            for k, v in attrs.items():
                assert _eq(got[k], v), (path, k, got[k], v)
        assert isinstance(f.attrs["nsteps"], np.int64) and isinstance(f.attrs["time"], np.float64)
        assert isinstance(f.attrs["solver"], str)
        for path, v in exp["data"].items():
            d = f[path]
            assert _eq(d[()], v), path
        assert f["state/density/data"].shape == (8, 6)
        assert f["state/density/data"].dtype == np.float64
        assert f["state/density/data"][2:4, 1].shape == (2,)
        assert f["misc/i32"].dtype == np.int32 and f["misc/f32"].dtype == np.float32
        assert f["BC/hse"][()] is np.True_ or f["BC/hse"][()] == True   # noqa: E712
        assert "state/energy/data" in f and "state/nope" not in f
        with pytest.raises(KeyError):
            f["state/nope"]
        with pytest.raises(OSError):
            f.create_group("x")


def _write_sample(fn, rng):
    exp = {}
    with h5pure.File(fn, "w") as f:
        f.attrs["solver"] = "advection"
        f.attrs["time"] = 0.25
        f.attrs["nsteps"] = 12
        g = f.create_group("grid")
        g.attrs["nx"] = 16
        st = f.create_group("state")
        for n in range(21):                      # > 8 links: several symbol-table nodes
            gv = st.create_group(f"var{n:02d}")
            d = rng.random((5, 3 + n))
            gv.create_dataset("data", data=d)
            gv.attrs["xlb"] = "periodic"
            exp[f"state/var{n:02d}/data"] = d
        rp = f.create_group("runtime parameters")
        for n in range(150):                     # a 10 KB object header
            rp.attrs[f"sec.p{n}"] = [n, n / 3.0, f"v{n}"][n % 3]
        m = f.create_group("misc")
        m.create_dataset("ints", data=np.arange(6).reshape(2, 3))
        m.create_dataset("flag", data=True)
        m.create_dataset("label", data="héllo")
        m.create_dataset("scalar", data=2.5)
        m.attrs["vec"] = np.array([1.0, 2.0])
        m.attrs["uni"] = "αβγ"
        m.attrs["empty"] = ""
        m.attrs["b"] = np.bool_(True)
        f.create_dataset("a/b/c", data=np.ones((2, 2), dtype=np.float32))
        f.create_group("empty")
        with pytest.raises(ValueError):
            f.create_group("grid")
    return exp


def test_write_read_roundtrip(tmp_path):
    fn = str(tmp_path / "w.h5")
    exp = _write_sample(fn, np.random.default_rng(3))
    with h5pure.File(fn) as f:
        assert list(f) == ["a", "empty", "grid", "misc", "runtime parameters", "state"]
        assert f.attrs["solver"] == "advection" and f.attrs["time"] == 0.25
        assert f.attrs["nsteps"] == 12 and f["grid"].attrs["nx"] == 16
        assert list(f["state"]) == [f"var{n:02d}" for n in range(21)]
        for k, d in exp.items():
            assert np.array_equal(f[k][...], d)
        rp = f["runtime parameters"].attrs
        assert len(rp) == 150 and rp["sec.p4"] == 4 / 3.0 and rp["sec.p5"] == "v5" and rp["sec.p6"] == 6
        m = f["misc"]
        assert np.array_equal(m["ints"][()], np.arange(6).reshape(2, 3)) and m["ints"].dtype == np.int64
        assert m["flag"][()] == True and m["label"][()] == "héllo" and m["scalar"][()] == 2.5   # noqa: E712
        assert np.array_equal(m.attrs["vec"], [1.0, 2.0]) and m.attrs["uni"] == "αβγ"
        assert m.attrs["empty"] == "" and m.attrs["b"] == True   # noqa: E712
        assert f["a/b/c"].dtype == np.float32 and len(f["empty"]) == 0


def test_not_hdf5_and_unsupported(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file" * 64)
    with pytest.raises(OSError):
        h5pure.File(str(p))
    sb2 = bytearray(h5pure.SIG + bytes(88))
    sb2[8] = 2
    p.write_bytes(bytes(sb2))
    with pytest.raises(NotImplementedError):
        h5pure.File(str(p))


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no interpreter with the real h5py here")
def test_real_h5py_reads_what_h5pure_writes(tmp_path):
    fn = str(tmp_path / "w.h5")
    _write_sample(fn, np.random.default_rng(3))
    code = r"""
import sys, json, h5py, numpy as np
f = h5py.File(sys.argv[1], "r")
out = {"top": sorted(f), "solver": f.attrs["solver"], "nsteps": int(f.attrs["nsteps"]),
       "time": float(f.attrs["time"]), "nstate": len(f["state"]),
       "sum": float(sum(f["state"][k]["data"][()].sum() for k in f["state"])),
       "xlb": f["state/var20"].attrs["xlb"], "nrp": len(f["runtime parameters"].attrs),
       "p5": f["runtime parameters"].attrs["sec.p5"], "p4": float(f["runtime parameters"].attrs["sec.p4"]),
       "uni": f["misc"].attrs["uni"], "flag": bool(f["misc/flag"][()]),
       "flag_dtype": str(f["misc/flag"].dtype), "ints": f["misc/ints"][()].tolist(),
       "label": f["misc/label"][()].decode("utf-8"), "c": str(f["a/b/c"].dtype)}
# libhdf5 accepts the file for modification too (heap / B-tree consistency)
f.close()
with h5py.File(sys.argv[1], "a") as g:
    for n in range(20):
        g["empty"].create_group("g%d" % n)
    g["misc"].attrs["more"] = 5
print(json.dumps(out))
"""
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    r = subprocess.run([CONDA, "-c", code, fn], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    rng = np.random.default_rng(3)
    s = sum(rng.random((5, 3 + n)).sum() for n in range(21))
    assert out["top"] == ["a", "empty", "grid", "misc", "runtime parameters", "state"]
    assert out["solver"] == "advection" and out["nsteps"] == 12 and out["time"] == 0.25
    assert out["nstate"] == 21 and abs(out["sum"] - s) < 1e-12 and out["xlb"] == "periodic"
    assert out["nrp"] == 150 and out["p5"] == "v5" and out["p4"] == 4 / 3.0
    assert out["uni"] == "αβγ" and out["flag"] is True and out["flag_dtype"] == "bool"
    assert out["ints"] == [[0, 1, 2], [3, 4, 5]] and out["label"] == "héllo" and out["c"] == "float32"
    with h5pure.File(fn) as f:      # ... and the file libhdf5 modified is still readable here
        assert len(f["empty"]) == 20 and f["misc"].attrs["more"] == 5


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reads_pyro_shipped_benchmarks():
    """the regression benchmarks pyro ships (pyro/*/tests/*.h5) open through
    io_pyro.read without h5py, like `pyro_sim.py --compare_benchmark` does"""
    os.environ["PYRO_H5PURE"] = "1"
    try:
        from pyro2_amd.util import io_pyro
        s = io_pyro.read(REF + "/advection/tests/smooth_0040")
        assert s.n == 40 and s.cc_data.grid.nx == 32 and s.cc_data.names == ["density"]
        d = s.cc_data.get_var("density").v()
        assert d.shape == (32, 32) and 1.0 < d.max() < 2.0
        s = io_pyro.read(REF + "/compressible/tests/sod_x_0076.h5")
        assert s.n == 76 and s.cc_data.get_aux("gamma") == 1.4
        assert s.cc_data.BCs["y-momentum"].ylb == "reflect-odd"
    finally:
        del os.environ["PYRO_H5PURE"]
