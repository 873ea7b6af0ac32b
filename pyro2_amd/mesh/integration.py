"""Runge-Kutta integration of a CellCenterData2d on the device, API of
pyro/mesh/integration.py:1-113 (RK2, TVD2, TVD3, RK4).

The stage states and the increments k_s are device states; building a stage
start or the final update is one pyrohip_state_lincomb launch (clone of the
start incl. ghost cells, then interior += dt*a[s][j]*k_j in the reference's
order of accumulation)."""
import numpy as np

from .. import device

a = {"RK2": np.array([[0.0, 0.0], [0.5, 0.0]]),
     "TVD2": np.array([[0.0, 0.0], [1.0, 0.0]]),
     "TVD3": np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.25, 0.25, 0.0]]),
     "RK4": np.array([[0.0, 0.0, 0.0, 0.0], [0.5, 0.0, 0.0, 0.0], [0.0, 0.5, 0.0, 0.0],
                      [0.0, 0.0, 1.0, 0.0]])}
b = {"RK2": np.array([0.0, 1.0]), "TVD2": np.array([0.5, 0.5]),
     "TVD3": np.array([1. / 6., 1. / 6., 2. / 3.]),
     "RK4": np.array([1. / 6., 1. / 3., 1. / 3., 1. / 6.])}
c = {"RK2": np.array([0.0, 0.5]), "TVD2": np.array([0.0, 1.0]),
     "TVD3": np.array([0.0, 1.0, 0.5]), "RK4": np.array([0.0, 0.5, 0.5, 1.0])}


class RKIntegrator:
    """integrates the DeviceState behind a CellCenterData2d.  Unlike the
    reference, the increments are written by the solver directly into slot s of
    `self.k` (a device state with nvar * nstages planes): store_increment only
    records that stage s is done."""

    def __init__(self, t, dt, method="RK4"):
        self.method, self.t, self.dt = method, t, dt
        self.start = None
        self.k = None
        self.stage = None
        self.done = 0

    def nstages(self):
        return len(b[self.method])

    def set_start(self, start, scratch=None):
        """start: DeviceState; scratch: (stage DeviceState, k DeviceState) to
        reuse between steps"""
        self.start = start
        if scratch is None:
            rows = [list(r) for r in start.bc]
            scratch = (device.DeviceState(start.ctx, start.nx, start.ny, start.ng, rows),
                       # the increments have no ghost cells to fill
                       device.DeviceState(start.ctx, start.nx, start.ny, start.ng,
                                          [["outflow"] * 4] * (len(rows) * self.nstages())))
        self.stage, self.k = scratch
        return scratch

    def get_stage_start(self, istage):
        if istage == 0:
            return self.start
        coefs = [self.dt * a[self.method][istage, s] for s in range(istage)]
        self.stage.lincomb(self.start, self.k, coefs)
        return self.stage

    def store_increment(self, istage, k_stage=None):
        self.done = istage + 1

    def compute_final_update(self):
        coefs = [self.dt * b[self.method][s] for s in range(self.nstages())]
        self.start.lincomb(self.start, self.k, coefs)
        return self.start

    def __str__(self):
        return f"integration method: {self.method}; number of stages: {self.nstages()}"
