"""diffusion.Simulation with the call surface of
pyro/diffusion/simulation.py:13-122.  One step = ghost fill, right-hand side
f = phi + (dt/2) k L(phi), a Helmholtz solve (alpha = 1, beta = (dt/2) k) with
the multigrid V-cycle, copy-back -- all on the device; phi never leaves HBM
inside Pyro.run_sim."""
import numpy as np

from .. import device
from ..mesh import patch
from ..simulation_null import NullSimulation, bc_setup, grid_setup
from ..util import msg


class Simulation(NullSimulation):
    def initialize(self):
        my_grid = grid_setup(self.rp, ng=1)
        if my_grid.nx != my_grid.ny:
            msg.fail("need nx = ny for diffusion problems")
        n = int(np.log(my_grid.nx) / np.log(2.0))
        if 2**n != my_grid.nx:
            msg.fail("grid needs to be a power of 2")
        bc, _, _ = bc_setup(self.rp)
        for b in bc.sides():
            if b not in ("periodic", "neumann", "dirichlet"):
                msg.fail("invalid BC")
        my_data = patch.CellCenterData2d(my_grid)
        my_data.register_var("phi", bc)
        my_data.create()
        self.cc_data = my_data
        self._mg = None
        self._mg_key = None
        self.problem_func(self.cc_data, self.rp)

    def method_compute_timestep(self):
        """explicit diffusion limit times driver.cfl (the scheme is implicit,
        cfl > 1 is fine), diffusion/simulation.py:53-70"""
        cfl = self.rp.get_param("driver.cfl")
        k = self.rp.get_param("diffusion.k")
        g = self.cc_data.grid
        self.dt = cfl * min(g.dx**2 / k, g.dy**2 / k)

    def _solver(self, beta):
        """the reference builds a new MG object every step
        (diffusion/simulation.py:92-101); here the device hierarchy is reused
        while alpha/beta and the BCs stay the same"""
        cc, g = self.cc_data, self.cc_data.grid
        key = (beta, cc.BCs["phi"].sides())
        if self._mg is None or self._mg_key != key:
            self._mg = device.DeviceMG(cc.ctx, g.nx, xmin=g.xmin, xmax=g.xmax, ymin=g.ymin,
                                       ymax=g.ymax, bcs=cc.BCs["phi"].sides(), alpha=1.0,
                                       beta=beta, nsmooth=10, nsmooth_bottom=50)
            self._mg_key = key
        return self._mg

    def evolve(self):
        tm = self.tc.timer("evolve")
        tm.begin()
        cc = self.cc_data
        cc.fill_BC_all()
        k = self.rp.get_param("diffusion.k")
        beta = 0.5 * self.dt * k
        mg = self._solver(beta)
        st = cc.device_state()
        n = cc.names.index("phi")
        mg.set_rhs_cn(st, n, beta)               # f = phi + (dt/2) k L phi, ||f||
        mg.zero(mg.nlevels - 1, 0)               # initial guess: zeros
        self.mg_cycles, self.mg_residual, _ = mg.solve(rtol=1.e-10)
        mg.copy_solution(st, n)
        cc.device_modified()
        cc.t += self.dt
        self.n += 1
        tm.end()

    def dovis(self):
        import matplotlib.pyplot as plt
        plt.clf()
        phi = self.cc_data.get_var("phi")
        g = self.cc_data.grid
        img = plt.imshow(np.transpose(phi.v()), interpolation="nearest", origin="lower",
                         extent=[g.xmin, g.xmax, g.ymin, g.ymax], cmap=self.cm)
        plt.xlabel("x")
        plt.ylabel("y")
        plt.title("phi")
        plt.colorbar(img)
        plt.figtext(0.05, 0.0125, f"t = {self.cc_data.t:10.5f}")
        plt.pause(0.001)
        plt.draw()
