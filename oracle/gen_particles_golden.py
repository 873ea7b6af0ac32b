"""tests/golden/particles.npz: the reference's tracer particles
(pyro/particles/particles.py) RUN on fixed velocity fields, plus the particle
records of the advection regression benchmark smooth_0040.h5.

    cd /tmp && MPLBACKEND=Agg PYTHONPATH=/root/repo/oracle/shim:/root/reference \
      /opt/conda/bin/python3.9 /root/repo/oracle/gen_particles_golden.py

Test infrastructure only; nothing of the reference is copied, it is called.
"""
import os
import tempfile

import h5py
import numpy as np

os.chdir(tempfile.mkdtemp())

import pyro.mesh.boundary as bnd                 # noqa: E402
from pyro.mesh import patch                      # noqa: E402
from pyro.particles import particles             # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
REF = "/root/reference/pyro"


def fields(g, kind):
    x, y = g.x2d, g.y2d
    u, v = g.scratch_array(), g.scratch_array()
    if kind == "swirl":
        u[:, :] = -np.sin(np.pi * x) ** 2 * np.sin(2 * np.pi * y) + 0.3
        v[:, :] = np.sin(np.pi * y) ** 2 * np.sin(2 * np.pi * x) - 0.2
    else:
        u[:, :] = 1.0 + 0.5 * y
        v[:, :] = -0.75 + 0.25 * x
    return u, v


def main():
    out = {}
    cases = [("per", "swirl", ["periodic"] * 4, 36, 0.021, 25),
             ("refl", "shear", ["reflect-even", "reflect-odd", "dirichlet", "reflect-even"], 49, 0.04, 30),
             ("out", "shear", ["outflow", "neumann", "outflow", "outflow"], 30, 0.05, 12)]
    for tag, kind, b, npart, dt, nsteps in cases:
        g = patch.Grid2d(24, 16, ng=4, xmin=0.0, xmax=1.5, ymin=-0.5, ymax=0.5)
        d = patch.CellCenterData2d(g)
        bc = bnd.BC(xlb=b[0], xrb=b[1], ylb=b[2], yrb=b[3])
        d.register_var("density", bc)
        d.create()
        u, v = fields(g, kind)
        ps = particles.Particles(d, bc, npart, "grid")
        out[f"{tag}_init0"] = ps.get_init_positions()
        hist, ihist, counts = [], [], []
        for _ in range(nsteps):
            ps.update_particles(dt, u, v)
            hist.append(ps.get_positions().reshape(-1, 2))
            ihist.append(ps.get_init_positions().reshape(-1, 2))
            counts.append(ps.n_particles)
        out[f"{tag}_counts"] = np.array(counts)
        out[f"{tag}_pos"] = np.concatenate(hist)          # ragged: split by counts
        out[f"{tag}_init"] = np.concatenate(ihist)
        out[f"{tag}_meta"] = np.array([npart, dt, nsteps])
        print(tag, counts[0], "->", counts[-1])
    with h5py.File(REF + "/advection/tests/smooth_0040.h5", "r") as f:
        out["smooth40_pos"] = f["particles/particle_positions"][()]
        out["smooth40_init"] = f["particles/init_particle_positions"][()]
    np.savez_compressed(os.path.join(OUT, "particles.npz"), **out)


if __name__ == "__main__":
    main()
