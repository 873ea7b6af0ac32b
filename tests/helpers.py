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


# Butcher tableaus of pyro/mesh/integration.py:27-60 (a, b)
RK_TABLEAU = {
    "RK2": ([[0.0, 0.0], [0.5, 0.0]], [0.0, 1.0]),
    "TVD2": ([[0.0, 0.0], [1.0, 0.0]], [0.5, 0.5]),
    "TVD3": ([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.25, 0.25, 0.0]], [1. / 6., 1. / 6., 2. / 3.]),
    "RK4": ([[0.0, 0.0, 0.0, 0.0], [0.5, 0.0, 0.0, 0.0], [0.0, 0.5, 0.0, 0.0],
             [0.0, 0.0, 1.0, 0.0]], [1. / 6., 1. / 3., 1. / 3., 1. / 6.]),
}


def oracle_rk_step(U, P, bcs, dt, method, ambient=(0.0,) * 4):
    """compressible_rk Simulation.evolve (simulation.py:58-95) with the
    RKIntegrator of mesh/integration.py:63-113 on the oracle"""
    a, b = RK_TABLEAU[method]
    ng = P.ng
    I = (slice(ng, -ng), slice(ng, -ng))
    ks = []
    for s in range(len(b)):
        if s == 0:
            y = U                      # the state itself: filled and floored in place
        else:
            y = U.copy()
            for j in range(s):
                y[I] += dt * a[s][j] * ks[j][I]
        orc.comp_fill_bc(y, P.nx, P.ny, P.ng, bcs, P.gamma, P.grav, P.dy, ambient)
        rc, k = orc.comp_rk_rhs(y, P)
        assert rc == 0
        ks.append(k)
    for s in range(len(b)):
        U[I] += dt * b[s] * ks[s][I]


def oracle_rk_run(ic, meta, bcs, nsteps, method, f0=0.01, mx=2.0, **over):
    P, cfl = meta_to_params(meta, bcs, **over)
    U = np.nan_to_num(np.ascontiguousarray(ic, dtype=np.float64))
    pol = DtPolicy(1.e30, f0, mx)
    dts = []
    for _ in range(nsteps):
        orc.comp_fill_bc(U, P.nx, P.ny, P.ng, bcs, P.gamma, P.grav, P.dy)
        dt = pol(orc.comp_rk_dt(U, P.nx, P.ny, P.ng, P.dx, P.dy, P.gamma, cfl))
        oracle_rk_step(U, P, bcs, dt, method)
        pol.advance(dt)
        dts.append(dt)
    return U, np.array(dts)
