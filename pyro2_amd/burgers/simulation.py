"""burgers.Simulation with the call surface of pyro/burgers/simulation.py:
14-187.  evolve() = pyrohip_bg_step (limited slopes, edge states, transverse
terms, Riemann/upwind fluxes and the conservative update on the device);
the CFL step needs max|u|, max|v| over the whole array: two device
min/max reductions."""
import numpy as np

from ..mesh import patch
from ..simulation_null import NullSimulation, bc_setup, grid_setup


class Simulation(NullSimulation):
    def initialize(self):
        my_grid = grid_setup(self.rp, ng=4)
        my_data = patch.CellCenterData2d(my_grid)
        bc = bc_setup(self.rp)[0]
        my_data.register_var("x-velocity", bc)
        my_data.register_var("y-velocity", bc)
        my_data.create()
        self.cc_data = my_data
        self.setup_particles(bc)
        self.problem_func(self.cc_data, self.rp)

    def _max_abs(self, name):
        cc = self.cc_data
        lo, hi = cc.device_state().minmax(cc.names.index(name), buf=cc.grid.ng)
        return max(-lo, hi)

    def method_compute_timestep(self):
        """cfl * min(dx / max|u|, dy / max|v|) over the whole array
        (burgers/simulation.py:37-51)"""
        cfl = self.rp.get_param("driver.cfl")
        g = self.cc_data.grid
        xtmp = g.dx / max(self._max_abs("x-velocity"), self.SMALL)
        ytmp = g.dy / max(self._max_abs("y-velocity"), self.SMALL)
        self.dt = cfl * min(xtmp, ytmp)

    def evolve(self):
        tm = self.tc.timer("evolve")
        tm.begin()
        cc, g = self.cc_data, self.cc_data.grid
        st = cc.device_state()
        st.bg_step(cc.names.index("x-velocity"), cc.names.index("y-velocity"), g.dx, g.dy,
                   self.dt, self.rp.get_param("advection.limiter"))
        cc.device_modified()
        self.advance_particles()         # burgers/simulation.py:128-133 (updated u, v)
        cc.t += self.dt
        self.n += 1
        tm.end()

    def dovis(self):
        import matplotlib.pyplot as plt
        plt.clf()
        g = self.cc_data.grid
        _, axes = plt.subplots(nrows=1, ncols=2, num=1, clear=True)
        for ax, name in zip(axes, ("x-velocity", "y-velocity")):
            img = ax.imshow(np.transpose(self.cc_data.get_var(name).v()), interpolation="nearest",
                            origin="lower", extent=[g.xmin, g.xmax, g.ymin, g.ymax], cmap=self.cm)
            ax.set_xlabel("x")
            ax.set_ylabel("y")
            ax.set_title(name)
            plt.colorbar(img, ax=ax)
        plt.figtext(0.05, 0.0125, f"t = {self.cc_data.t:10.5f}")
        plt.pause(0.001)
        plt.draw()
