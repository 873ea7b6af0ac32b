"""Writes tests/golden/h5py_written.h5 with the real h5py / libhdf5 (run under
/opt/conda/bin/python3.9, the interpreter of this container that has h5py):
a file in pyro's output layout (pyro/simulation_null.py:270-290,
pyro/mesh/patch.py:750-788, pyro/util/runparams.py write_params) plus the
corner cases the pure-Python reader pyro2_amd/util/h5pure.py has to handle
(groups with more than 8 links -> several symbol-table nodes, objects with
many attributes -> object header continuation blocks, bool / int / string
datasets).  The expected content is stored next to it as JSON.

Test infrastructure: only tests/ read these files.
"""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")


def main():
    rng = np.random.default_rng(20260926)
    exp = {"attrs": {}, "data": {}}

    def setattr_(obj, path, k, v):
        obj.attrs[k] = v
        exp["attrs"].setdefault(path, {})[k] = v.tolist() if isinstance(v, np.ndarray) else v

    fn = os.path.join(OUT, "h5py_written.h5")
    with h5py.File(fn, "w") as f:
        for k, v in (("solver", "compressible"), ("problem", "sedov"), ("time", 0.0125),
                     ("nsteps", 17), ("dt", 1.25e-3), ("dt_old", 1.0e-3)):
            setattr_(f, "/", k, v)
        g = f.create_group("grid")
        for k, v in (("nx", 8), ("ny", 6), ("ng", 4), ("xmin", 0.0), ("xmax", 1.0),
                     ("ymin", -0.5), ("ymax", 0.25)):
            setattr_(g, "grid", k, v)
        a = f.create_group("aux")
        setattr_(a, "aux", "gamma", 1.4)
        setattr_(a, "aux", "grav", -2.0)
        st = f.create_group("state")
        for n, name in enumerate(["density", "energy", "x-momentum", "y-momentum"] +
                                 [f"scalar{k:02d}" for k in range(14)]):
            gv = st.create_group(name)
            d = rng.random((8, 6)) - 0.25 * n
            gv.create_dataset("data", data=d)
            exp["data"][f"state/{name}/data"] = d.tolist()
            for t, b in zip(("xlb", "xrb", "ylb", "yrb"),
                            ("outflow", "outflow", "reflect-even", "reflect-odd")):
                setattr_(gv, f"state/{name}", t, b)
        rp = f.create_group("runtime parameters")
        for n in range(120):
            v = [n, n / 7.0, f"value-{n}"][n % 3]
            setattr_(rp, "runtime parameters", f"section{n % 5}.param_{n}", v)
        bc = f.create_group("BC")
        bc.create_dataset("hse", data=True)
        exp["data"]["BC/hse"] = True
        m = f.create_group("misc")
        m.create_dataset("i32", data=np.arange(12, dtype=np.int32).reshape(3, 4))
        exp["data"]["misc/i32"] = np.arange(12).reshape(3, 4).tolist()
        m.create_dataset("f32", data=np.arange(5, dtype=np.float32) / 4)
        exp["data"]["misc/f32"] = (np.arange(5) / 4).tolist()
        m.create_dataset("scalar", data=3.5)
        exp["data"]["misc/scalar"] = 3.5
        setattr_(m, "misc", "vec", np.array([1.0, 2.5, -3.0]))
        setattr_(m, "misc", "ivec", np.array([1, 2, 3]))
        setattr_(m, "misc", "unicode", "αβγ pyro")
        setattr_(m, "misc", "empty", "")
        f.create_group("empty")
    json.dump(exp, open(os.path.join(OUT, "h5py_written.json"), "w"), indent=0)
    print("wrote", fn, os.path.getsize(fn), "bytes")


if __name__ == "__main__":
    main()
