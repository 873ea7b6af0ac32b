"""Shared drivers for the oracle (CPU checker) used by several tests."""
import numpy as np

from oracle import orc


def meta_to_params(meta, bcs, **over):
    nx, ny, ng, dx, dy, gamma, lim, flat, z0, z1, delta, cvisc, grav, cfl = meta
    kw = dict(gamma=gamma, limiter=int(lim), use_flattening=int(flat), z0=z0,
              z1=z1, delta=delta, cvisc=cvisc, grav=grav, bcs=tuple(bcs))
    kw.update(over)
    return orc.comp_params(int(nx), int(ny), int(ng), dx, dy, **kw), cfl


class DtPolicy:
    """driver dt policy, pyro/simulation_null.py:222-244"""

    def __init__(self, tmax, init_tstep_factor=0.01, max_dt_change=2.0,
                 fix_dt=-1.0):
        self.tmax, self.f0, self.mx, self.fix = tmax, init_tstep_factor, max_dt_change, fix_dt
        self.n = 0
        self.t = 0.0
        self.dt_old = -1.e33

    def __call__(self, dt_method):
        if self.fix > 0.0:
            dt = self.fix
        else:
            dt = dt_method
            if self.n == 0:
                dt = self.f0 * dt
            else:
                dt = min(self.mx * self.dt_old, dt)
            self.dt_old = dt
        if self.t + dt > self.tmax:
            dt = self.tmax - self.t
        return dt

    def advance(self, dt):
        self.t += dt
        self.n += 1


def oracle_comp_run(ic, meta, bcs, tmax, max_steps, init_tstep_factor=0.01,
                    max_dt_change=2.0, ambient=(0.0,) * 4, **over):
    """run the C oracle like Pyro.run_sim (pyro_sim.py:219-256)"""
    P, cfl = meta_to_params(meta, bcs, **over)
    U = np.ascontiguousarray(ic, dtype=np.float64).copy()
    pol = DtPolicy(tmax, init_tstep_factor, max_dt_change)
    dts = []
    while not (pol.t >= tmax or pol.n >= max_steps):
        orc.comp_fill_bc(U, P.nx, P.ny, P.ng, bcs, P.gamma, P.grav, P.dy, ambient)
        dtm = orc.comp_dt(U, P.nx, P.ny, P.ng, P.dx, P.dy, P.gamma, cfl)
        dt = pol(dtm)
        rc, _ = orc.comp_step(U, P, dt)
        assert rc == 0
        pol.advance(dt)
        dts.append(dt)
    return U, np.array(dts), pol.t
