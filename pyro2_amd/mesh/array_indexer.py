"""ArrayIndexer: ndarray subclass with the shifted-view algebra of
pyro/mesh/array_indexer.py:29-148 (v / ip / jp / ip_jp / lap / norm / copy ...).

Host-side convenience only -- views into host copies of the data.  The ghost
fill of the hot path runs on the device (CellCenterData2d.fill_BC);
ArrayIndexer.fill_ghost on a stand-alone host array round-trips through the
same device kernel.
"""
import numbers

import numpy as np


def _buf_split(b):
    """scalar | (lo, hi) | (xlo, xhi, ylo, yhi) -> four widths"""
    try:
        n = len(b)
    except TypeError:
        return b, b, b, b
    if n == 4:
        return tuple(b)
    if n == 2:
        return b[0], b[1], b[0], b[1]
    raise ValueError("buf must be a scalar, a pair or a 4-tuple")


class ArrayIndexer(np.ndarray):
    def __new__(cls, d, grid=None):
        obj = np.asarray(d).view(cls)
        obj.g = grid
        obj.c = len(d.shape)
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self.g = getattr(obj, "g", None)
        self.c = getattr(obj, "c", None)

    # ---- shifted views ------------------------------------------------
    def ip_jp(self, ishift, jshift, buf=0, n=0, s=1):
        """view of the valid region grown by buf, shifted by (ishift, jshift),
        stride s, component n"""
        bxlo, bxhi, bylo, byhi = _buf_split(buf)
        g = self.g
        si = slice(g.ilo - bxlo + ishift, g.ihi + 1 + bxhi + ishift, s)
        sj = slice(g.jlo - bylo + jshift, g.jhi + 1 + byhi + jshift, s)
        if self.ndim == 2:
            return np.asarray(self[si, sj])
        return np.asarray(self[si, sj, n])

    def v(self, buf=0, n=0, s=1):
        return self.ip_jp(0, 0, buf=buf, n=n, s=s)

    def ip(self, shift, buf=0, n=0, s=1):
        return self.ip_jp(shift, 0, buf=buf, n=n, s=s)

    def jp(self, shift, buf=0, n=0, s=1):
        return self.ip_jp(0, shift, buf=buf, n=n, s=s)

    def lap(self, n=0, buf=0):
        """5-point Laplacian (array_indexer.py:92-96)"""
        c = self.v(n=n, buf=buf)
        return (self.ip(-1, n=n, buf=buf) - 2 * c + self.ip(1, n=n, buf=buf)) / self.g.dx**2 + \
               (self.jp(-1, n=n, buf=buf) - 2 * c + self.jp(1, n=n, buf=buf)) / self.g.dy**2

    def norm(self, n=0):
        """sqrt(dx dy sum a^2) over the valid region (array_indexer.py:98-111)"""
        a = self if self.ndim == 2 else self[:, :, n]
        g = self.g
        inner = np.asarray(a)[g.ilo:g.ihi + 1, g.jlo:g.jhi + 1]
        return np.sqrt(g.dx * g.dy * np.sum((inner**2).flat))

    def copy(self, order="C"):
        return ArrayIndexer(np.asarray(self).copy(order=order), grid=self.g)

    def is_symmetric(self, nodal=False, tol=1.e-14, asymmetric=False):
        g = self.g
        sgn = -1 if asymmetric else 1
        half = g.nx // 2
        if nodal:
            left = self[g.ilo:g.ilo + half + 1, g.jlo:g.jhi + 1]
            right = self[g.ilo + half:g.ihi + 2, g.jlo:g.jhi + 1]
        else:
            left = self[g.ilo:g.ilo + half, g.jlo:g.jhi + 1]
            right = self[g.ilo + half:g.ihi + 1, g.jlo:g.jhi + 1]
        return abs(np.asarray(left) - sgn * np.flipud(np.asarray(right))).max() < tol

    def is_asymmetric(self, nodal=False, tol=1.e-14):
        return self.is_symmetric(nodal=nodal, tol=tol, asymmetric=True)

    # ---- ghost cells --------------------------------------------------
    def fill_ghost(self, n=0, bc=None):
        """fill the ghost cells of component n according to bc
        (array_indexer.py:150-274) with the device kernel"""
        from .. import device
        from .patch import _fill_host_array
        _fill_host_array(device.Context.default(), self, n, bc)

    def pretty_print(self, n=0, fmt=None, show_ghost=True):
        """print a small array with ghost cells in red, j downwards"""
        if fmt is None:
            if issubclass(self.dtype.type, numbers.Integral):
                fmt = "%4d"
            elif self.dtype == np.float64:
                fmt = "%10.5g"
            else:
                raise ValueError("ERROR: dtype not supported")
        g = self.g
        if show_ghost:
            ir, jr = range(0, g.qx), range(0, g.qy)
        else:
            ir, jr = range(g.ilo, g.ihi + 1), range(g.jlo, g.jhi + 1)
        for j in reversed(jr):
            row = ""
            for i in ir:
                val = self[i, j] if self.ndim == 2 else self[i, j, n]
                ghost = not (g.ilo <= i <= g.ihi and g.jlo <= j <= g.jhi)
                row += ("\033[31m" + fmt % val + "\033[0m") if ghost else fmt % val
            print(row + " ")
        print("\n         ^ y\n         |\n         +---> x\n        ")
