"""Output files of the package: pyro's HDF5 layout (simulation_null.py:270-290,
patch.py:750-788), SURVEY.md 8 row f3.

`open_file(name, mode)` returns
  * an h5py.File when h5py is importable,
  * otherwise a util/h5pure.File: a pure Python + NumPy reader / writer of the
    HDF5 subset pyro uses.  Files it writes are real HDF5 (libhdf5 / h5py and
    therefore pyro's own io_pyro / compare.py / plot.py read them); it reads
    the files pyro writes, including the regression benchmarks pyro ships.
Both expose the same small surface (groups, attrs, create_dataset, item
access).  The `NpzFile` class below is the container earlier versions of this
package wrote when h5py was missing ("<base>.pyro.npz"); it is kept so that
those files stay readable.
"""
import os

import numpy as np


def have_h5py():
    try:
        import h5py  # noqa: F401
        return True
    except ImportError:
        return False


def resolve(filename, for_write):
    """full file name for a base name (with or without extension)"""
    if filename.endswith(".h5") or filename.endswith(".pyro.npz"):
        return filename
    if for_write:
        return filename + ".h5"
    for ext in (".h5", ".pyro.npz"):
        if os.path.exists(filename + ext):
            return filename + ext
    return filename + ".h5"


def open_file(filename, mode="r"):
    name = resolve(filename, for_write=mode != "r")
    if name.endswith(".pyro.npz"):
        return NpzFile(name, mode)
    if have_h5py() and not os.environ.get("PYRO_H5PURE"):
        import h5py
        return h5py.File(name, mode)
    from . import h5pure
    return h5pure.File(name, mode)


class _Attrs:
    def __init__(self, root, path):
        self._root, self._path = root, path

    def _key(self, k):
        return f"a:{self._path}@{k}"

    def __setitem__(self, k, v):
        self._root._store[self._key(k)] = np.asarray(v)

    def __getitem__(self, k):
        v = self._root._store[self._key(k)]
        return v.item() if v.ndim == 0 else v

    def get(self, k, default=None):
        return self[k] if self._key(k) in self._root._store else default

    def __contains__(self, k):
        return self._key(k) in self._root._store

    def __iter__(self):
        pre = f"a:{self._path}@"
        return iter(sorted(k[len(pre):] for k in self._root._store if k.startswith(pre)))

    def keys(self):
        return list(self)

    def items(self):
        return [(k, self[k]) for k in self]


class _Group:
    def __init__(self, root, path):
        self._root, self._path = root, path
        self.attrs = _Attrs(root, path)

    def _child(self, name):
        return f"{self._path}/{name}" if self._path else name

    def create_group(self, name):
        p = self._child(name)
        self._root._store["g:" + p] = np.zeros(0)
        return _Group(self._root, p)

    def create_dataset(self, name, data=None):
        self._root._store["d:" + self._child(name)] = np.array(data)
        return self._root._store["d:" + self._child(name)]

    def _children(self):
        pre = self._path + "/" if self._path else ""
        out = set()
        for k in self._root._store:
            kind, p = k[0], k[2:]
            if kind == "a":
                p = p.split("@", 1)[0]
            if p.startswith(pre) and len(p) > len(pre):
                out.add(p[len(pre):].split("/", 1)[0])
        return sorted(out)

    def __iter__(self):
        return iter(self._children())

    def keys(self):
        return self._children()

    def __contains__(self, name):
        return name.split("/", 1)[0] in self._children() and \
            ("d:" + self._child(name) in self._root._store or
             any(c == name.split("/")[-1] for c in _Group(
                 self._root, "/".join(self._child(name).split("/")[:-1]))._children()))

    def __getitem__(self, name):
        p = self._child(name)
        if "d:" + p in self._root._store:
            return self._root._store["d:" + p]
        g = _Group(self._root, p)
        if "g:" + p in self._root._store or g._children() or list(g.attrs):
            return g
        raise KeyError(name)


class NpzFile(_Group):
    def __init__(self, filename, mode="r"):
        self.filename, self.mode = filename, mode
        self._store = {}
        if mode == "r":
            with np.load(filename, allow_pickle=False) as z:
                self._store = {k: z[k] for k in z.files}
        super().__init__(self, "")

    def close(self):
        if self.mode != "r":
            with open(self.filename, "wb") as fh:
                np.savez(fh, **self._store)
            self.mode = "r"

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
