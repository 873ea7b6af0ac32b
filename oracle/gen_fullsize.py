"""Full-size fixtures from the C ORACLE (not the reference, which cannot run
these sizes in test time): BASELINE config 2, compressible Sedov 4096^2
(inputs.sedov physics), state after NSTEPS steps from the deterministic IC.
The state does not fit a fixture: a 64x64 lattice of samples, row / column
sums per variable and the dt sequence are kept (tests/golden/
comp_sedov_4096_samples.npz).  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_fullsize.py        # ~10 min on one core
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import oracle_comp_run    # noqa: E402
from sedov_ic import sedov_ic          # noqa: E402

NX, NSTEPS = 4096, 25


def main():
    ic, meta, bcs = sedov_ic(NX)
    t0 = time.time()
    U, dts, t = oracle_comp_run(ic, meta, bcs, 0.1, NSTEPS)
    print("oracle", NX, NSTEPS, "steps:", time.time() - t0, "s")
    I = U[4:-4, 4:-4]
    step = NX // 64
    out = os.path.join(ROOT, "tests", "golden", "comp_sedov_4096_samples.npz")
    np.savez_compressed(out, samples=I[::step, ::step].copy(), row_sums=I.sum(axis=1),
                        col_sums=I.sum(axis=0), dts=dts, t=np.array(t), nsteps=np.array(NSTEPS),
                        umax=np.abs(I).max(axis=(0, 1)))
    print("wrote", out, os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    main()
