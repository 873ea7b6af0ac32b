"""HDF5 reader / writer in pure Python + NumPy for the subset of the format
pyro's output files use ("h5py-free path").

pyro writes its output with h5py (pyro/simulation_null.py:270-290,
pyro/mesh/patch.py:750-788, pyro/util/runparams.py `write_params`): nested
groups, scalar attributes (int64 / float64 / variable-length UTF-8 strings)
and contiguous float64 datasets.  This module reads such files -- including
the regression benchmarks shipped with pyro -- and writes files that h5py /
libhdf5 (and therefore pyro's own io_pyro.read, compare.py, plot.py) read
back, without h5py being installed.

On-disk structures (HDF5 File Format Specification, version 1.1 subset, the
"earliest" format h5py writes by default):
    superblock v0 -> root symbol-table entry
    groups:   object header v1 + symbol-table message -> B-tree v1 node
              ("TREE") -> symbol-table nodes ("SNOD") + local heap ("HEAP")
    datasets: object header v1 with dataspace / datatype / fill-value /
              layout v3 (contiguous or compact) messages
    attributes: attribute messages v1-v3 in the object header (+ continuation
              blocks); variable-length strings in global heap collections
              ("GCOL")
Not supported (clear NotImplementedError): superblock >= 2, object header v2,
chunked / filtered datasets, dense attribute or link storage, compound types.

The object surface is the part of h5py's the package uses: File(name, mode),
group[...], `in`, iteration, create_group, create_dataset(name, data=...),
.attrs (mapping), dataset[...] / .shape / .dtype.
"""
import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
GROUP_LEAF_K = 4           # symbol-table node holds up to 2K entries
GROUP_INTERNAL_K = 16      # B-tree node holds up to 2K children


def _pad8(n):
    return (n + 7) & ~7


# ---------------------------------------------------------------------------
# reading
# ---------------------------------------------------------------------------
class _Raw:
    def __init__(self, buf):
        self.b = buf

    def u8(self, p):
        return self.b[p]

    def u16(self, p):
        return struct.unpack_from("<H", self.b, p)[0]

    def u32(self, p):
        return struct.unpack_from("<I", self.b, p)[0]

    def u64(self, p):
        return struct.unpack_from("<Q", self.b, p)[0]

    def cstr(self, p):
        e = self.b.index(b"\0", p)
        return bytes(self.b[p:e]).decode("utf-8")


class _DType:
    """decoded datatype message"""

    def __init__(self, kind, size, np_dtype=None, base=None, is_bool=False, cset="utf-8"):
        self.kind, self.size, self.np_dtype, self.base = kind, size, np_dtype, base
        self.is_bool, self.cset = is_bool, cset


def _parse_dtype(raw, p):
    """-> (_DType, bytes consumed)"""
    b0 = raw.u8(p)
    cls, ver = b0 & 0x0F, b0 >> 4
    bits = (raw.u8(p + 1), raw.u8(p + 2), raw.u8(p + 3))
    size = raw.u32(p + 4)
    q = p + 8
    if cls == 0:        # fixed point
        order = ">" if bits[0] & 1 else "<"
        signed = bool(bits[0] & 8)
        return _DType("int", size, np.dtype(f"{order}{'i' if signed else 'u'}{size}")), 8 + 4
    if cls == 1:        # floating point
        order = ">" if bits[0] & 1 else "<"
        return _DType("float", size, np.dtype(f"{order}f{size}")), 8 + 12
    if cls == 3:        # fixed-length string
        return _DType("string", size, np.dtype(f"S{size}"),
                      cset="utf-8" if (bits[0] >> 4) & 0xF else "ascii"), 8
    if cls == 9:        # variable length
        base, n = _parse_dtype(raw, q)
        is_str = (bits[0] & 0x0F) == 1
        return _DType("vlen_str" if is_str else "vlen", size, base=base), 8 + n
    if cls == 8:        # enumeration (h5py stores bool as enum of int8 FALSE/TRUE)
        nmemb = bits[0] | (bits[1] << 8)
        base, n = _parse_dtype(raw, q)
        q += n
        names = []
        for _ in range(nmemb):
            s = raw.cstr(q)
            names.append(s)
            q += _pad8(len(s) + 1) if ver < 3 else len(s) + 1
        q += nmemb * base.size
        is_bool = sorted(names) == ["FALSE", "TRUE"]
        return _DType("enum", size, base.np_dtype, base=base, is_bool=is_bool), q - p
    if cls == 7:        # reference
        return _DType("ref", size, np.dtype(f"V{size}")), 8
    raise NotImplementedError(f"h5pure: HDF5 datatype class {cls} is not supported")


def _parse_dspace(raw, p):
    ver, rank, flags = raw.u8(p), raw.u8(p + 1), raw.u8(p + 2)
    if ver == 1:
        q = p + 8
    elif ver == 2:
        if raw.u8(p + 3) == 2:      # null dataspace
            return None
        q = p + 4
    else:
        raise NotImplementedError(f"h5pure: dataspace message version {ver}")
    return tuple(raw.u64(q + 8 * k) for k in range(rank))


class _RFile:
    """parsed view of a file image"""

    def __init__(self, buf):
        self.raw = raw = _Raw(buf)
        base = 0
        while bytes(buf[base:base + 8]) != SIG:      # user block: 512, 1024, ...
            base = 512 if base == 0 else base * 2
            if base + 8 > len(buf):
                raise OSError("h5pure: not an HDF5 file (signature not found)")
        ver = raw.u8(base + 8)
        if ver > 1:
            raise NotImplementedError(
                f"h5pure: superblock version {ver} (files written with libver='latest') "
                "is not supported")
        if raw.u8(base + 13) != 8 or raw.u8(base + 14) != 8:
            raise NotImplementedError("h5pure: only 8-byte offsets / lengths")
        p = base + 24 + (4 if ver == 1 else 0)
        self.base = raw.u64(p)
        root_entry = p + 32
        self.root_addr = raw.u64(root_entry + 8)
        self._gcol = {}

    # -- object headers ------------------------------------------------------
    def messages(self, addr):
        raw = self.raw
        addr += self.base
        ver = raw.u8(addr)
        if ver != 1:
            raise NotImplementedError("h5pure: object header version 2 is not supported")
        nmsgs, hsize = raw.u16(addr + 2), raw.u32(addr + 8)
        chunks = [(addr + 16, hsize)]
        out = []
        while chunks and len(out) < nmsgs:
            p, sz = chunks.pop(0)
            end = p + sz
            while p + 8 <= end and len(out) < nmsgs:
                t, s, fl = raw.u16(p), raw.u16(p + 2), raw.u8(p + 4)
                if t == 0x10:
                    chunks.append((self.base + raw.u64(p + 8), raw.u64(p + 16)))
                out.append((t, fl, p + 8, s))
                p += 8 + s
        return out

    def kind(self, addr):
        types = {t for (t, _, _, _) in self.messages(addr)}
        if 0x11 in types:
            return "group"
        if 0x08 in types:
            return "dataset"
        if 0x06 in types or 0x02 in types:
            raise NotImplementedError("h5pure: new-style (link message) groups are not supported")
        return "group"     # a group without any link yet cannot occur in v1 files

    # -- groups ----------------------------------------------------------------
    def links(self, addr):
        """ordered {name: object header address} of a group"""
        raw = self.raw
        for (t, fl, p, s) in self.messages(addr):
            if t == 0x11:
                btree, heap = self.base + raw.u64(p), self.base + raw.u64(p + 8)
                break
        else:
            return {}
        if bytes(raw.b[heap:heap + 4]) != b"HEAP":
            raise OSError("h5pure: bad local heap signature")
        hdata = self.base + raw.u64(heap + 24)
        out = {}

        def walk(node):
            sig = bytes(raw.b[node:node + 4])
            if sig == b"TREE":
                n = raw.u16(node + 6)
                for k in range(n):
                    walk(self.base + raw.u64(node + 24 + 8 + 16 * k))
            elif sig == b"SNOD":
                n = raw.u16(node + 6)
                for k in range(n):
                    e = node + 8 + 40 * k
                    out[raw.cstr(hdata + raw.u64(e))] = raw.u64(e + 8)
            else:
                raise OSError("h5pure: bad group node signature")
        walk(btree)
        return out

    # -- values ------------------------------------------------------------------
    def _gheap_obj(self, addr, idx):
        raw = self.raw
        addr += self.base
        if addr not in self._gcol:
            if bytes(raw.b[addr:addr + 4]) != b"GCOL":
                raise OSError("h5pure: bad global heap signature")
            size = raw.u64(addr + 8)
            objs, p = {}, addr + 16
            while p + 16 <= addr + size:
                i, osz = raw.u16(p), raw.u64(p + 8)
                if i == 0:
                    break
                objs[i] = (p + 16, osz)
                p += 16 + _pad8(osz)
            self._gcol[addr] = objs
        p, n = self._gcol[addr][idx]
        return bytes(raw.b[p:p + n])

    def decode(self, dt, shape, p):
        """value of `shape` (None: null space, (): scalar) stored at byte p"""
        raw = self.raw
        if shape is None:
            return None
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if dt.kind == "vlen_str":
            vals = []
            for k in range(n):
                q = p + 16 * k
                ln, a, i = raw.u32(q), raw.u64(q + 4), raw.u32(q + 12)
                vals.append(self._gheap_obj(a, i)[:ln].decode("utf-8") if ln else "")
            if shape == ():
                return vals[0]
            return np.array(vals, dtype=object).reshape(shape)
        if dt.kind == "vlen":
            raise NotImplementedError("h5pure: variable-length sequences are not supported")
        a = np.frombuffer(raw.b, dtype=dt.np_dtype, count=n, offset=p).reshape(shape)
        if dt.kind == "enum" and dt.is_bool:
            a = a.astype(np.bool_)
        elif dt.np_dtype.byteorder == ">":
            a = a.astype(dt.np_dtype.newbyteorder("<"))
        if dt.kind == "string":
            if shape == ():
                return bytes(a[()]).rstrip(b"\0").decode(dt.cset)
            return a.copy()
        if shape == ():
            return a[()]
        return a.copy()

    def attributes(self, addr):
        raw = self.raw
        out = {}
        for (t, fl, p, s) in self.messages(addr):
            if t == 0x15:
                raise NotImplementedError("h5pure: dense attribute storage is not supported")
            if t != 0x0C:
                continue
            if fl & 2:
                raise NotImplementedError("h5pure: shared attribute messages are not supported")
            ver = raw.u8(p)
            nsz, tsz, ssz = raw.u16(p + 2), raw.u16(p + 4), raw.u16(p + 6)
            if ver == 1:
                q = p + 8
                name = raw.cstr(q); q += _pad8(nsz)
                dt, _ = _parse_dtype(raw, q); q += _pad8(tsz)
                shape = _parse_dspace(raw, q); q += _pad8(ssz)
            elif ver in (2, 3):
                if raw.u8(p + 1) & 3:
                    raise NotImplementedError("h5pure: shared datatype / dataspace in attribute")
                q = p + 8 + (1 if ver == 3 else 0)
                name = raw.cstr(q); q += nsz
                dt, _ = _parse_dtype(raw, q); q += tsz
                shape = _parse_dspace(raw, q); q += ssz
            else:
                raise NotImplementedError(f"h5pure: attribute message version {ver}")
            out[name] = self.decode(dt, shape, q)
        return out

    def dataset(self, addr):
        """(dtype, shape, loader)"""
        raw = self.raw
        dt = shape = None
        layout = None
        for (t, fl, p, s) in self.messages(addr):
            if fl & 2 and t in (1, 3, 8):
                raise NotImplementedError("h5pure: shared (committed) messages are not supported")
            if t == 0x01:
                shape = _parse_dspace(raw, p)
            elif t == 0x03:
                dt, _ = _parse_dtype(raw, p)
            elif t == 0x08:
                ver = raw.u8(p)
                if ver != 3:
                    raise NotImplementedError(f"h5pure: data layout message version {ver}")
                cls = raw.u8(p + 1)
                if cls == 1:
                    layout = ("contiguous", raw.u64(p + 2), raw.u64(p + 10))
                elif cls == 0:
                    layout = ("compact", p + 4, raw.u16(p + 2))
                else:
                    raise NotImplementedError(
                        "h5pure: chunked / filtered datasets are not supported")
            elif t == 0x0B:
                raise NotImplementedError("h5pure: filtered datasets are not supported")
        if dt is None or layout is None:
            raise OSError("h5pure: incomplete dataset object header")

        def load():
            if layout[0] == "contiguous":
                if layout[1] == UNDEF:      # never written: fill value (0)
                    if dt.kind == "vlen_str":
                        return "" if shape == () else np.full(shape, "", dtype=object)
                    z = np.zeros(shape if shape else (), dtype=dt.np_dtype)
                    return z[()] if shape == () else z
                return self.decode(dt, shape, self.base + layout[1])
            return self.decode(dt, shape, layout[1])
        return dt, shape, load


# ---------------------------------------------------------------------------
# in-memory tree shared by reading and writing
# ---------------------------------------------------------------------------
class Attrs:
    """attribute mapping of a group / dataset (h5py's AttributeManager)"""

    def __init__(self, owner):
        self._o = owner
        self._d = None

    def _load(self):
        if self._d is None:
            o = self._o
            self._d = o._file._r.attributes(o._addr) if o._addr is not None else {}
        return self._d

    def __getitem__(self, k):
        return self._load()[k]

    def __setitem__(self, k, v):
        self._o._file._check_writable()
        self._load()[str(k)] = v

    def __contains__(self, k):
        return k in self._load()

    def __iter__(self):
        return iter(self._load())

    def __len__(self):
        return len(self._load())

    def get(self, k, default=None):
        return self._load().get(k, default)

    def keys(self):
        return self._load().keys()

    def values(self):
        return self._load().values()

    def items(self):
        return self._load().items()


class Dataset:
    def __init__(self, file, name, addr=None, data=None):
        self._file, self.name, self._addr = file, name, addr
        self.attrs = Attrs(self)
        if addr is not None:
            self._dt, self.shape, self._loader = file._r.dataset(addr)
            self._data = None
        else:
            self._data = data
            self.shape = data.shape if isinstance(data, np.ndarray) else ()

    def _value(self):
        if self._data is None:
            self._data = self._loader()
        return self._data

    @property
    def dtype(self):
        v = self._value()
        return v.dtype if hasattr(v, "dtype") else np.dtype(object)

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1

    def __getitem__(self, idx):
        v = self._value()
        if isinstance(v, np.ndarray):
            return v[idx]
        if idx == () or idx is Ellipsis:
            return v
        raise IndexError("scalar dataset: index with [()]")

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self._value())
        return a.astype(dtype) if dtype is not None else a

    def __len__(self):
        return self.shape[0]

    def __repr__(self):
        return f'<h5pure dataset "{self.name}": shape {self.shape}>'


class Group:
    def __init__(self, file, name, addr=None):
        self._file, self.name, self._addr = file, name, addr
        self.attrs = Attrs(self)
        self._kids = None

    def _children(self):
        if self._kids is None:
            self._kids = {}
            if self._addr is not None:
                r = self._file._r
                for n, a in r.links(self._addr).items():
                    path = (self.name.rstrip("/") + "/" + n)
                    self._kids[n] = (Group(self._file, path, a) if r.kind(a) == "group"
                                     else Dataset(self._file, path, a))
        return self._kids

    def _walk(self, path, create=False):
        node = self._file if path.startswith("/") else self
        parts = [s for s in path.split("/") if s]
        for i, s in enumerate(parts):
            if not isinstance(node, Group):
                raise KeyError(path)
            kids = node._children()
            if s not in kids:
                if not create:
                    raise KeyError(f"'{path}' does not exist in {self.name}")
                kids[s] = Group(self._file, node.name.rstrip("/") + "/" + s)
            node = kids[s]
        return node

    def __getitem__(self, path):
        return self._walk(path)

    def __contains__(self, path):
        try:
            self._walk(path)
            return True
        except KeyError:
            return False

    def __iter__(self):
        return iter(sorted(self._children()))    # h5py iterates in name order

    def __len__(self):
        return len(self._children())

    def keys(self):
        return sorted(self._children())

    def values(self):
        return [self._children()[k] for k in self.keys()]

    def items(self):
        return [(k, self._children()[k]) for k in self.keys()]

    def get(self, path, default=None):
        try:
            return self._walk(path)
        except KeyError:
            return default

    def visititems(self, fn):
        def rec(g, prefix):
            for k, v in g.items():
                r = fn(prefix + k, v)
                if r is not None:
                    return r
                if isinstance(v, Group):
                    r = rec(v, prefix + k + "/")
                    if r is not None:
                        return r
            return None
        return rec(self, "")

    def _split(self, path):
        parts = [s for s in path.split("/") if s]
        if not parts:
            raise ValueError("empty name")
        root = "/" if path.startswith("/") else ""
        parent = self._walk(root + "/".join(parts[:-1]), create=True) if len(parts) > 1 else \
            (self._file if root else self)
        return parent, parts[-1]

    def create_group(self, path):
        self._file._check_writable()
        parent, leaf = self._split(path)
        if leaf in parent._children():
            raise ValueError(f"unable to create group (name already exists): {path}")
        g = Group(self._file, parent.name.rstrip("/") + "/" + leaf)
        parent._children()[leaf] = g
        return g

    def require_group(self, path):
        return self[path] if path in self else self.create_group(path)

    def create_dataset(self, path, shape=None, dtype=None, data=None):
        self._file._check_writable()
        parent, leaf = self._split(path)
        if leaf in parent._children():
            raise ValueError(f"unable to create dataset (name already exists): {path}")
        if data is None:
            data = np.zeros(shape if shape is not None else (), dtype=dtype or np.float64)
        if not isinstance(data, str):
            data = np.array(data, dtype=dtype)      # a copy, like h5py
            if data.dtype.kind in "US":
                data = data.astype(object) if data.ndim else str(data[()])
        d = Dataset(self._file, parent.name.rstrip("/") + "/" + leaf, data=data)
        parent._children()[leaf] = d
        return d

    def __repr__(self):
        return f'<h5pure group "{self.name}" ({len(self)} members)>'


class File(Group):
    """h5py.File work-alike: mode "r" parses an existing file lazily, mode
    "w" builds the tree in memory and serialises it on close()"""

    def __init__(self, filename, mode="r"):
        if mode not in ("r", "w"):
            raise ValueError("h5pure.File: mode must be 'r' or 'w'")
        self.filename, self.mode = filename, mode
        self._open = True
        if mode == "r":
            with open(filename, "rb") as fh:
                self._r = _RFile(fh.read())
            super().__init__(self, "/", self._r.root_addr)
        else:
            self._r = None
            open(filename, "wb").close()     # fail early on an unwritable path
            super().__init__(self, "/", None)

    def _check_writable(self):
        if self.mode != "w" or not self._open:
            raise OSError("h5pure: file is not open for writing")

    def close(self):
        if self._open and self.mode == "w":
            img = _Writer().serialise(self)
            with open(self.filename, "wb") as fh:
                fh.write(img)
        self._open = False

    def flush(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __repr__(self):
        return f'<h5pure file "{self.filename}" (mode {self.mode})>'


# ---------------------------------------------------------------------------
# writing
# ---------------------------------------------------------------------------
_DT_F64 = struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 0x3F, 0x00, 8, 0, 64, 52, 11, 0, 52, 1023)
_DT_F32 = struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 0x1F, 0x00, 4, 0, 32, 23, 8, 0, 23, 127)
_DT_VSTR = (struct.pack("<BBBBI", 0x19, 0x01, 0x01, 0x00, 16) +      # vlen, string, UTF-8
            struct.pack("<BBBBI", 0x13, 0x10, 0x00, 0x00, 1))        # base: 1-byte UTF-8 string


def _dt_int(size, signed):
    return struct.pack("<BBBBIHH", 0x10, 0x08 if signed else 0x00, 0, 0, size, 0, 8 * size)


def _dt_bool():
    # enum over int8 with members FALSE = 0, TRUE = 1 (what h5py writes)
    base = _dt_int(1, True)
    names = b"FALSE\0\0\0" + b"TRUE\0\0\0\0"
    return struct.pack("<BBBBI", 0x18, 0x02, 0x00, 0x00, 1) + base + names + b"\x00\x01"


def _dspace(shape):
    return struct.pack("<BBBB4x", 1, len(shape), 0, 0) + b"".join(
        struct.pack("<Q", int(n)) for n in shape)


class _Writer:
    def __init__(self):
        self.img = bytearray(96)        # superblock written last
        self.strings = []               # (patch position, utf-8 bytes) of vlen strings

    def alloc(self, data):
        while len(self.img) % 8:
            self.img.append(0)
        addr = len(self.img)
        self.img += data
        return addr

    # -- values -> (datatype message, dataspace message, payload) ------------
    def encode(self, v):
        if isinstance(v, (bytes, np.bytes_)):
            v = v.decode("utf-8")
        if isinstance(v, str):
            return _DT_VSTR, _dspace(()), [v]
        a = np.asarray(v)
        if a.dtype == object or a.dtype.kind in "US":
            strs = [s.decode("utf-8") if isinstance(s, bytes) else str(s) for s in a.ravel()]
            return _DT_VSTR, _dspace(a.shape), strs
        if a.dtype.kind == "b":
            return _dt_bool(), _dspace(a.shape), a.astype(np.int8).tobytes()
        if a.dtype.kind == "f":
            if a.dtype.itemsize == 4:
                return _DT_F32, _dspace(a.shape), a.astype("<f4").tobytes()
            return _DT_F64, _dspace(a.shape), a.astype("<f8").tobytes()
        if a.dtype.kind in "iu":
            sz = a.dtype.itemsize
            return (_dt_int(sz, a.dtype.kind == "i"), _dspace(a.shape),
                    np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<")).tobytes())
        raise TypeError(f"h5pure: cannot store values of type {a.dtype}")

    def payload(self, pl):
        """bytes of a payload; variable-length strings are 16-byte references
        patched once the global heap is laid out"""
        if isinstance(pl, bytes):
            return pl, []
        fix = []
        for k, s in enumerate(pl):
            fix.append((16 * k, s.encode("utf-8")))
        return bytes(16 * len(pl)), fix

    @staticmethod
    def msg(mtype, body, flags=0):
        body = body + bytes(_pad8(len(body)) - len(body))
        if len(body) > 0xFFF8:
            raise ValueError("h5pure: header message larger than 64 KiB")
        return struct.pack("<HHB3x", mtype, len(body), flags) + body

    def attr_msgs(self, attrs):
        """-> list of (message bytes, [(offset of a vlen reference in it, string)])"""
        out = []
        for name, v in attrs.items():
            dt, ds, pl = self.encode(v)
            nm = name.encode("utf-8") + b"\0"
            data, fix = self.payload(pl)
            head = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds))
            body = head + nm + bytes(_pad8(len(nm)) - len(nm)) + \
                dt + bytes(_pad8(len(dt)) - len(dt)) + ds + bytes(_pad8(len(ds)) - len(ds))
            off = 8 + len(body)         # message header + body so far
            out.append((self.msg(0x0C, body + data), [(off + o, s) for o, s in fix]))
        return out

    def object_header(self, msgs):
        """msgs: list of (bytes, fixups) -> address"""
        total = sum(len(m) for m, _ in msgs)
        head = struct.pack("<BBHII4x", 1, 0, len(msgs), 1, total)
        addr = self.alloc(head)
        for m, fix in msgs:
            p = len(self.img)
            self.img += m
            for off, s in fix:
                self.strings.append((p + off, s))
        return addr

    def write_dataset(self, d):
        v = d._value()
        dt, ds, pl = self.encode(v)
        data, fix = self.payload(pl)
        if len(data):
            daddr = self.alloc(data)
            for off, s in fix:
                self.strings.append((daddr + off, s))
        else:
            daddr = UNDEF
        msgs = [(self.msg(0x01, ds), []), (self.msg(0x03, dt, flags=1), []),
                (self.msg(0x05, struct.pack("<BBBB", 2, 2, 2, 0)), []),     # fill value v2: none
                (self.msg(0x08, struct.pack("<BBQQ", 3, 1, daddr, len(data))), [])]
        return self.object_header(msgs + self.attr_msgs(dict(d.attrs.items())))

    def write_group(self, g):
        kids = g._children()
        names = sorted(kids, key=lambda s: s.encode("utf-8"))
        addrs = {}
        for n in names:
            c = kids[n]
            addrs[n] = self.write_group(c) if isinstance(c, Group) else self.write_dataset(c)
        # local heap: offset 0 = "", then the names
        heap = bytearray(8)
        noff = {}
        for n in names:
            noff[n] = len(heap)
            b = n.encode("utf-8") + b"\0"
            heap += b + bytes(_pad8(len(b)) - len(b))
        hdata = self.alloc(bytes(heap))
        haddr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), 1, hdata))
        # symbol table nodes of up to 2K entries under ONE B-tree node
        per = 2 * GROUP_LEAF_K
        leaves = [names[k:k + per] for k in range(0, len(names), per)]
        if len(leaves) > 2 * GROUP_INTERNAL_K:
            raise ValueError("h5pure: more than 256 links in one group")
        snods = []
        for leaf in leaves:
            body = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(leaf)))
            for n in leaf:
                c = kids[n]
                if isinstance(c, Group):
                    body += struct.pack("<QQII", noff[n], addrs[n], 1, 0) + \
                        struct.pack("<QQ", *c._stab)
                else:
                    body += struct.pack("<QQII16x", noff[n], addrs[n], 0, 0)
            body += bytes(8 + 40 * per - len(body))
            snods.append(self.alloc(bytes(body)))
        node = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, len(leaves), UNDEF, UNDEF))
        node += struct.pack("<Q", 0)                     # key 0: the empty string
        for leaf, a in zip(leaves, snods):
            node += struct.pack("<QQ", a, noff[leaf[-1]])   # child, key = its largest name
        node += bytes(24 + 8 * (2 * GROUP_INTERNAL_K + 1) + 8 * 2 * GROUP_INTERNAL_K - len(node))
        baddr = self.alloc(bytes(node))
        g._stab = (baddr, haddr)
        msgs = [(self.msg(0x11, struct.pack("<QQ", baddr, haddr)), [])]
        return self.object_header(msgs + self.attr_msgs(dict(g.attrs.items())))

    def global_heap(self):
        """lay the variable-length strings out in global heap collections and
        patch the (length, collection address, index) references"""
        todo = list(self.strings)
        while todo:
            chunk, used = [], 16
            while todo and len(chunk) < 0xFFF0:
                need = 16 + _pad8(len(todo[0][1]))
                if chunk and used + need + 16 > 4096:
                    break
                chunk.append(todo.pop(0))
                used += need
            size = max(4096, _pad8(used + 16))
            col = bytearray(b"GCOL" + struct.pack("<B3xQ", 1, size))
            refs = []
            for k, (pos, s) in enumerate(chunk, start=1):
                col += struct.pack("<HH4xQ", k, 0, len(s)) + s + bytes(_pad8(len(s)) - len(s))
                refs.append((pos, len(s), k))
            free = size - len(col)
            col += struct.pack("<HH4xQ", 0, 0, free) + bytes(free - 16)
            caddr = self.alloc(bytes(col))
            for pos, ln, k in refs:
                if ln:
                    struct.pack_into("<IQI", self.img, pos, ln, caddr, k)

    def serialise(self, f):
        root = self.write_group(f)
        self.global_heap()
        while len(self.img) % 8:
            self.img.append(0)
        sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, GROUP_LEAF_K,
                               GROUP_INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.img), UNDEF)
        sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", *f._stab)
        assert len(sb) == 96
        self.img[0:96] = sb
        return bytes(self.img)
