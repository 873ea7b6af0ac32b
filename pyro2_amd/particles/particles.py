"""Massless tracer particles advected with the cell-centred velocity, API of
pyro/particles/particles.py:11-364 (`Particle`, `Particles`).

pyro keeps a dict {initial position: Particle} and walks it in Python; here the
particles are two (n, 2) arrays (current and initial positions) and every step
is a handful of NumPy expressions over all particles -- the same arithmetic,
expression by expression, so positions agree with pyro's bit for bit:

  * midpoint update (particles.py:213-257): velocity at the old position,
    half step, velocity there, full step from the old position;
  * bilinear interpolation from the four surrounding cell centres incl. the
    reference's index rule `int(x_idx) + 1` into the buf=1 view
    (particles.py:46-86);
  * boundaries (particles.py:259-327): outflow drops a particle, periodic
    wraps, reflecting walls mirror; pyro rebuilds its dict with popitem(), so
    the ORDER of the particles reverses at every update -- kept, because the
    order is what `get_positions()` and the output file show.

Particles are a host-side diagnostic of O(100) points.  The velocity field is
passed in by the solver (constant for advection) or taken from the state's
derived "velocity" (downloaded from the device once per step, only when
particles are enabled).
"""
import numpy as np

from ..util import msg


class Particle:
    """one particle (particles.py:11-44); what `Particles.particles` hands out"""

    def __init__(self, x, y, u=0, v=0):
        self.x, self.y, self.u, self.v = x, y, u, v

    def pos(self):
        return np.array([self.x, self.y])

    def velocity(self):
        return np.array([self.u, self.v])

    def update(self, u, v, dt):
        self.u, self.v = u, v
        self.x += u * dt
        self.y += v * dt

    def interpolate_velocity(self, myg, u, v):
        uu, vv = _interpolate(myg, np.asarray(u), np.asarray(v),
                              np.array([self.x]), np.array([self.y]))
        return uu[0], vv[0]


def _interpolate(myg, u, v, x, y):
    """bilinear interpolation of the full (ghosted) arrays u, v to (x, y),
    particles.py:62-86"""
    x_idx = (x - myg.xmin) / myg.dx - 0.5
    y_idx = (y - myg.ymin) / myg.dy - 0.5
    x_frac = x_idx % 1
    y_frac = y_idx % 1
    # index into the buf=1 view == full-array index ilo - 1 + k
    i = np.trunc(x_idx).astype(np.int64) + 1 + myg.ilo - 1
    j = np.trunc(y_idx).astype(np.int64) + 1 + myg.jlo - 1

    def interp(a):
        return (1 - x_frac) * (1 - y_frac) * a[i, j] + \
            x_frac * (1 - y_frac) * a[i + 1, j] + \
            (1 - x_frac) * y_frac * a[i, j + 1] + \
            x_frac * y_frac * a[i + 1, j + 1]
    return interp(u), interp(v)


class Particles:
    def __init__(self, sim_data, bc, n_particles, particle_generator="grid",
                 pos_array=None, init_array=None):
        self.sim_data, self.bc = sim_data, bc
        self.pos = np.zeros((0, 2))
        self.init = np.zeros((0, 2))
        self.vel = np.zeros((0, 2))
        if n_particles <= 0:
            msg.fail(f"ERROR: n_particles = {n_particles} <= 0")
        if callable(particle_generator):          # {(x0, y0): Particle}
            d = particle_generator(n_particles)
            self.init = np.array([[k[0], k[1]] for k in d], dtype=np.float64).reshape(-1, 2)
            self.pos = np.array([[p.x, p.y] for p in d.values()], dtype=np.float64).reshape(-1, 2)
        elif particle_generator == "random":
            self.randomly_generate_particles(n_particles)
        elif particle_generator == "grid":
            self.grid_generate_particles(n_particles)
        elif particle_generator == "array":
            self.array_generate_particles(pos_array, init_array)
        else:
            msg.fail(f"ERROR: do not recognise particle generator {particle_generator}")
        self._dedupe()
        self.vel = np.zeros_like(self.pos)

    # pyro keys its dict by the initial position: a second particle with the
    # same initial position replaces the first (keeping the first one's slot)
    def _dedupe(self):
        if len(self.init) < 2:
            return
        slot, order = {}, []
        for n, k in enumerate(map(tuple, self.init)):
            if k not in slot:
                order.append(k)
            slot[k] = n
        if len(order) != len(self.init):
            keep = [slot[k] for k in order]
            self.init, self.pos = self.init[keep], self.pos[keep]

    @property
    def n_particles(self):
        return len(self.pos)

    @property
    def particles(self):
        """{(x0, y0): Particle} like pyro's attribute (a snapshot)"""
        return {(i[0], i[1]): Particle(p[0], p[1], w[0], w[1])
                for i, p, w in zip(self.init, self.pos, self.vel)}

    def randomly_generate_particles(self, n_particles):
        myg = self.sim_data.grid
        positions = np.random.rand(n_particles, 2)
        positions[:, 0] = positions[:, 0] * (myg.xmax - myg.xmin) + myg.xmin
        positions[:, 1] = positions[:, 1] * (myg.ymax - myg.ymin) + myg.ymin
        self.init, self.pos = positions.copy(), positions.copy()

    def grid_generate_particles(self, n_particles):
        """sqrt(n) x sqrt(n) particles at the centres of equal boxes
        (particles.py:163-187); x is the slow index"""
        sq = int(round(np.sqrt(n_particles)))
        if sq ** 2 != n_particles:
            msg.warning(f"WARNING: Changing number of particles from {n_particles} to {sq**2}")
        myg = self.sim_data.grid
        xs, step = np.linspace(myg.xmin, myg.xmax, num=sq, endpoint=False, retstep=True)
        xs += 0.5 * step
        ys, step = np.linspace(myg.ymin, myg.ymax, num=sq, endpoint=False, retstep=True)
        ys += 0.5 * step
        X, Y = np.meshgrid(xs, ys, indexing="ij")
        self.init = np.stack([X.ravel(), Y.ravel()], axis=1)
        self.pos = self.init.copy()

    def array_generate_particles(self, pos_array, init_array=None):
        if pos_array is None:
            msg.fail("ERROR: Array of particle positions has not been passed into "
                     "Particles constructor. Cannot generate particles.")
        self.pos = np.array(pos_array, dtype=np.float64).reshape(-1, 2)
        self.init = self.pos.copy() if init_array is None else \
            np.array(init_array, dtype=np.float64).reshape(-1, 2)

    def update_particles(self, dt, u=None, v=None):
        """midpoint update with the cell-centred velocity (particles.py:213-257)"""
        myg = self.sim_data.grid
        if (u is None) and (v is None):
            u, v = self.sim_data.get_var("velocity")
        elif u is None:
            u = self.sim_data.get_var("x-velocity")
        elif v is None:
            v = self.sim_data.get_var("y-velocity")
        u, v = np.asarray(u), np.asarray(v)
        x0, y0 = self.pos[:, 0].copy(), self.pos[:, 1].copy()
        u_vel, v_vel = _interpolate(myg, u, v, x0, y0)
        hdt = 0.5 * dt
        xh = x0 + u_vel * hdt
        yh = y0 + v_vel * hdt
        u_vel, v_vel = _interpolate(myg, u, v, xh, yh)
        self.pos = np.stack([x0 + u_vel * dt, y0 + v_vel * dt], axis=1)
        self.vel = np.stack([u_vel, v_vel], axis=1)
        self.enforce_particle_boundaries()

    def enforce_particle_boundaries(self):
        myg, bc = self.sim_data.grid, self.bc
        x, y = self.pos[:, 0].copy(), self.pos[:, 1].copy()
        keep = np.ones(len(x), dtype=bool)
        drop = ("outflow", "neumann")
        mirror = ("reflect-even", "reflect-odd", "dirichlet")

        def side(c, below, kind, lo, hi, tag):
            m = keep & ((c < lo) if below else (c > hi))
            if not m.any():
                return
            if kind in drop:
                keep[m] = False
            elif kind == "periodic":
                c[m] = (hi + c[m] - lo) if below else (lo + c[m] - hi)
            elif kind in mirror:
                c[m] = (2 * lo - c[m]) if below else (2 * hi - c[m])
            else:
                msg.fail(f"ERROR: {tag} = {kind} invalid BC for particles")
        side(x, True, bc.xlb, myg.xmin, myg.xmax, "xlb")
        side(x, False, bc.xrb, myg.xmin, myg.xmax, "xrb")
        side(y, True, bc.ylb, myg.ymin, myg.ymax, "ylb")
        side(y, False, bc.yrb, myg.ymin, myg.ymax, "yrb")
        # popitem() takes pyro's dict apart from the back: the order reverses
        sel = np.flatnonzero(keep)[::-1]
        self.pos = np.stack([x[sel], y[sel]], axis=1)
        self.init = self.init[sel]
        self.vel = self.vel[sel] if len(self.vel) == len(keep) else np.zeros_like(self.pos)

    def get_positions(self):
        return self.pos.copy()

    def get_init_positions(self):
        return self.init.copy()

    def write_particles(self, f):
        """group "particles" of the output file (particles.py:344-364)"""
        g = f.create_group("particles")
        g.create_dataset("init_particle_positions", data=self.get_init_positions())
        g.create_dataset("particle_positions", data=self.get_positions())
