#!/usr/bin/env python3
"""Zone-by-zone comparison of two CellCenterData2d objects / output files with
the surface of pyro/util/compare.py:22-91.  Note (kept from the reference):
np.allclose adds its default atol = 1e-8 to rtol*|b|;
pass atol=0 for a strict relative comparison."""
import sys

import numpy as np

errors = {"gridbad": "grids don't agree",
          "namesbad": "variable lists don't agree",
          "varerr": "one or more variables don't agree"}


def compare(data1, data2, rtol=1.e-12, atol=1.e-8):
    if not data1.grid == data2.grid:
        return "gridbad"
    if sorted(data1.names) != sorted(data2.names):
        return "namesbad"
    print(" \nvariable comparisons:")
    result = 0
    for name in data1.names:
        d1 = data1.get_var(name).v()
        d2 = data2.get_var(name).v()
        abs_err = np.max(np.abs(d1 - d2))
        if not np.any(d2 == 0):
            rel_err = np.max(np.abs(d1 - d2) / np.abs(d2))
            print(f"{name:20s} absolute error = {abs_err:10.10g}, relative error = {rel_err:10.10g}")
        else:
            print(f"{name:20s} absolute error = {abs_err:10.10g}")
        if not np.allclose(d1, d2, rtol=rtol, atol=atol):
            result = "varerr"
    return result


def main():
    from . import io_pyro as io
    if len(sys.argv) not in (3, 4):
        print("\n      usage: compare.py file1 file2 (rtol)\n")
        sys.exit(2)
    s1, s2 = io.read(sys.argv[1]), io.read(sys.argv[2])
    kw = {"rtol": float(sys.argv[3])} if len(sys.argv) == 4 else {}
    result = compare(s1.cc_data, s2.cc_data, **kw)
    print("SUCCESS: files agree" if result == 0 else "ERROR:  " + errors[result])


if __name__ == "__main__":
    main()
