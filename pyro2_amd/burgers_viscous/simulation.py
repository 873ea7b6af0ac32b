"""burgers_viscous.Simulation with the call surface of
pyro/burgers_viscous/simulation.py:7-89.

evolve() on the device:
  1. bgv_predict   limited slopes, edge states + eps dt / 2 L(U) (interface.py:
                   94-171), transverse terms, MAC velocities
  2. per component bgv_rhs (f = w + dt eps / 2 L(w) - dt A, A from the unsplit
                   fluxes; guess 0), MG solve to 1e-12, solution -> w
                   (interface.diffuse, interface.py:27-91)
The reference builds a new MG object for every solve; here one hierarchy per
set of boundary types is kept and its beta = dt eps / 2 is reset every step.
"""
import numpy as np

from .. import device
from ..burgers import Simulation as burgers_simulation
from ..util import msg

_MG_BC = {"periodic": "periodic", "outflow": "neumann", "neumann": "neumann",
          "dirichlet": "dirichlet", "reflect-even": "reflect-even", "reflect-odd": "dirichlet"}


class Simulation(burgers_simulation):
    def initialize(self):
        super().initialize()
        g = self.cc_data.grid
        if g.nx != g.ny or 2**int(round(np.log2(g.nx))) != g.nx:
            msg.fail("the multigrid solver needs nx = ny = 2^n")
        self._mgs = {}
        self.mg_cycles = (0, 0)

    def _mg(self, bcs, beta):
        g = self.cc_data.grid
        key = tuple(bcs)
        if key not in self._mgs:
            self._mgs[key] = device.DeviceMG(
                self.cc_data.ctx, g.nx, xmin=g.xmin, xmax=g.xmax, ymin=g.ymin, ymax=g.ymax,
                bcs=[_MG_BC[b] for b in bcs], alpha=1.0, beta=beta, nsmooth=10, nsmooth_bottom=50)
        mg = self._mgs[key]
        mg.set_helmholtz(1.0, beta)
        return mg

    def evolve(self):
        tm = self.tc.timer("evolve")
        tm.begin()
        cc, g = self.cc_data, self.cc_data.grid
        limiter = self.rp.get_param("advection.limiter")
        eps = self.rp.get_param("diffusion.eps")
        iu, iv = cc.names.index("x-velocity"), cc.names.index("y-velocity")
        st = cc.device_state()
        st.bgv_predict(iu, iv, g.dx, g.dy, self.dt, limiter, eps)
        cycles = []
        for comp, (iw, name) in enumerate(((iu, "x-velocity"), (iv, "y-velocity"))):
            mg = self._mg(cc.BCs[name].sides(), 0.5 * self.dt * eps)
            st.bgv_rhs(mg, iw, comp, g.dx, g.dy, self.dt, eps)
            cycles.append(mg.solve(rtol=1.e-12)[0])
            st.inc_visc_store(mg, iw)
        self.mg_cycles = tuple(cycles)
        cc.device_modified()
        self.advance_particles()         # burgers_viscous/simulation.py:80-85
        cc.t += self.dt
        self.n += 1
        tm.end()
