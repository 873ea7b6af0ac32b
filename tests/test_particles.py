"""pyro2_amd/particles: tracer particles, pinned on the reference's own
particles.py RUN on fixed velocity fields (tests/golden/particles.npz,
generator oracle/gen_particles_golden.py) -- positions, initial positions and
their ORDER, bit for bit -- and on the particle record of pyro's advection
regression benchmark smooth_0040.h5."""
import os

import numpy as np
import pytest

from pyro2_amd.mesh import boundary as bnd
from pyro2_amd.mesh import patch
from pyro2_amd.particles import particles

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "particles.npz"))


def _fields(g, kind):
    x, y = np.asarray(g.x2d), np.asarray(g.y2d)
    u, v = g.scratch_array(), g.scratch_array()
    if kind == "swirl":
        u[:, :] = -np.sin(np.pi * x) ** 2 * np.sin(2 * np.pi * y) + 0.3
        v[:, :] = np.sin(np.pi * y) ** 2 * np.sin(2 * np.pi * x) - 0.2
    else:
        u[:, :] = 1.0 + 0.5 * y
        v[:, :] = -0.75 + 0.25 * x
    return u, v


@pytest.mark.parametrize("tag,kind,b", [
    ("per", "swirl", ["periodic"] * 4),
    ("refl", "shear", ["reflect-even", "reflect-odd", "dirichlet", "reflect-even"]),
    ("out", "shear", ["outflow", "neumann", "outflow", "outflow"])])
def test_matches_reference_bit_for_bit(tag, kind, b):
    npart, dt, nsteps = GOLD[f"{tag}_meta"]
    g = patch.Grid2d(24, 16, ng=4, xmin=0.0, xmax=1.5, ymin=-0.5, ymax=0.5)
    d = patch.CellCenterData2d(g)
    bc = bnd.BC(xlb=b[0], xrb=b[1], ylb=b[2], yrb=b[3])
    d.register_var("density", bc)
    d.create()
    u, v = _fields(g, kind)
    ps = particles.Particles(d, bc, int(npart), "grid")
    assert np.array_equal(ps.get_init_positions(), GOLD[f"{tag}_init0"])
    counts = GOLD[f"{tag}_counts"]
    ends = np.cumsum(counts)
    for n in range(int(nsteps)):
        ps.update_particles(float(dt), u, v)
        sl = slice(ends[n] - counts[n], ends[n])
        assert ps.n_particles == counts[n]
        assert np.array_equal(ps.get_positions(), GOLD[f"{tag}_pos"][sl]), (tag, n)
        assert np.array_equal(ps.get_init_positions(), GOLD[f"{tag}_init"][sl]), (tag, n)
    if tag == "out":
        assert counts[-1] < counts[0]          # particles did leave the domain
    dct = ps.particles                          # pyro's dict view
    assert len(dct) == ps.n_particles
    k = next(iter(dct))
    assert np.array_equal(dct[k].pos(), ps.get_positions()[0])


def test_generators_and_single_particle():
    g = patch.Grid2d(8, 8, ng=2)
    d = patch.CellCenterData2d(g)
    bc = bnd.BC()
    d.register_var("a", bc)
    d.create()
    ps = particles.Particles(d, bc, 10, "random")
    assert ps.n_particles == 10 and ps.get_positions().min() >= 0 and ps.get_positions().max() <= 1
    ps = particles.Particles(d, bc, 3, "array", [[0.1, 0.2], [0.3, 0.4], [0.5, 0.6]])
    assert np.array_equal(ps.get_init_positions(), ps.get_positions())
    ps = particles.Particles(d, bc, 2, "array", [[0.1, 0.2], [0.3, 0.4]], [[0.0, 0.0], [1.0, 1.0]])
    assert np.array_equal(ps.get_init_positions(), [[0.0, 0.0], [1.0, 1.0]])
    ps = particles.Particles(d, bc, 1, lambda n: {(0.5, 0.5): particles.Particle(0.25, 0.75)})
    assert np.array_equal(ps.get_positions(), [[0.25, 0.75]])
    with pytest.raises(SystemExit):
        particles.Particles(d, bc, 0)
    with pytest.raises(SystemExit):
        particles.Particles(d, bc, 4, "nope")
    p = particles.Particle(0.4, 0.6)
    u = g.scratch_array() + 2.0
    v = g.scratch_array() - 1.0
    assert p.interpolate_velocity(g, u, v) == (2.0, -1.0)
    p.update(2.0, -1.0, 0.1)
    assert np.allclose(p.pos(), [0.6, 0.5]) and np.array_equal(p.velocity(), [2.0, -1.0])


def test_advection_smooth_benchmark_particles(dev, tmp_path, monkeypatch):
    """inputs.smooth carries 100 tracers; after the 40 steps of pyro's
    regression test their record equals the one in smooth_0040.h5, and it goes
    through the output file and io_pyro.read"""
    from pyro2_amd import device
    from pyro2_amd.pyro_sim import Pyro
    from pyro2_amd.util import io_pyro
    monkeypatch.setattr(device.Context, "_default", dev)
    monkeypatch.chdir(tmp_path)
    p = Pyro("advection")
    p.initialize_problem("smooth")
    assert p.sim.particles is not None and p.sim.particles.n_particles == 100
    p.run_sim()
    assert p.sim.n == 40
    assert np.array_equal(p.sim.particles.get_init_positions(), GOLD["smooth40_init"])
    np.testing.assert_allclose(p.sim.particles.get_positions(), GOLD["smooth40_pos"],
                               rtol=1e-13, atol=1e-15)
    p.sim.write("out_0040")
    s = io_pyro.read("out_0040")
    assert s.particles is not None and s.particles.n_particles == 100
    assert np.array_equal(s.particles.get_positions(), p.sim.particles.get_positions())
    assert np.array_equal(s.particles.get_init_positions(), GOLD["smooth40_init"])


def test_compressible_particles_follow_the_flow(dev, tmp_path, monkeypatch):
    """derived "velocity" of the compressible state moves the tracers without
    disturbing the device-resident run (same state as a run without them)"""
    from pyro2_amd import device
    from pyro2_amd.pyro_sim import Pyro
    monkeypatch.setattr(device.Context, "_default", dev)
    monkeypatch.chdir(tmp_path)
    d = {"mesh.nx": 32, "mesh.ny": 32, "driver.max_steps": 4, "sedov.r_init": 0.15}
    a = Pyro("compressible")
    a.initialize_problem("sedov", inputs_dict=dict(d))
    a.run_sim()
    b = Pyro("compressible")
    b.initialize_problem("sedov", inputs_dict=dict(d, **{"particles.do_particles": 1,
                                                          "particles.n_particles": 64,
                                                          "particles.particle_generator": "grid"}))
    b.run_sim()
    assert np.array_equal(np.asarray(a.sim.cc_data.data), np.asarray(b.sim.cc_data.data))
    ps = b.sim.particles
    assert ps.n_particles == 64
    moved = np.linalg.norm(ps.get_positions() - ps.get_init_positions(), axis=1)
    r0 = np.linalg.norm(ps.get_init_positions() - 0.5, axis=1)
    assert moved.max() > 0 and r0[np.argmax(moved)] < 0.3   # the edge of the hot region (r_init = 0.15) moves first
    m = moved > 0
    disp = ps.get_positions() - ps.get_init_positions()
    assert np.all(np.sum(disp[m] * (ps.get_init_positions()[m] - 0.5), axis=1) > 0)   # radially outwards
