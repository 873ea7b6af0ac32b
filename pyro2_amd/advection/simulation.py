"""advection.Simulation with the call surface of
pyro/advection/simulation.py:12-94; evolve() is one launch of the fused
LDS-tiled kernel (pyrohip_adv_step)."""
from ..mesh import patch
from ..simulation_null import NullSimulation, bc_setup, grid_setup


class Simulation(NullSimulation):
    # steps the driver hands over at once when it batches (a multiple of the two / three steps
    # a launch of the several-steps kernel takes)
    batch_steps = 96

    def initialize(self):
        """grid (ng = 4, advection/simulation.py:20), the single variable
        "density", then the problem's initial condition"""
        # (x-slabs with one process per GPU: the halo rows travel in cc_data.fill_BC_all)
        my_grid = grid_setup(self.rp, ng=4, decomposable=True)
        my_data = patch.CellCenterData2d(my_grid)
        bc = bc_setup(self.rp)[0]
        my_data.register_var("density", bc)
        my_data.create()
        # the step kernel does the ghost fill itself (index remap at load):
        # fill_BC_all() of the driver is deferred into it, one launch per step
        my_data.lazy_fill = True
        self.cc_data = my_data
        self.setup_particles(bc)         # advection/simulation.py:30-33
        self.problem_func(self.cc_data, self.rp)

    def method_compute_timestep(self):
        """closed-form advective CFL step (advection/simulation.py:38-54);
        nothing to reduce"""
        cfl = self.rp.get_param("driver.cfl")
        u = self.rp.get_param("advection.u")
        v = self.rp.get_param("advection.v")
        g = self.cc_data.grid
        xtmp = g.dx / max(abs(u), self.SMALL)
        ytmp = g.dy / max(abs(v), self.SMALL)
        self.dt = cfl * min(xtmp, ytmp)

    def _fast_math(self):
        """gpu.fast_math (default 1: the contracted build; 0: the bit-faithful audit build)"""
        try:
            return int(self.rp.get_param("gpu.fast_math"))
        except (KeyError, ValueError):
            return 1

    def evolve(self):
        """one time step of "density" on the device"""
        tm = self.tc.timer("evolve")
        tm.begin()
        g = self.cc_data.grid
        st = self.cc_data.device_state(fuse_fill=True)
        st.adv_step(self.cc_data.names.index("density"), g.dx, g.dy,
                    float(self.rp.get_param("advection.u")),
                    float(self.rp.get_param("advection.v")), float(self.dt),
                    int(self.rp.get_param("advection.limiter")),
                    fill=self.cc_data.take_pending_fill(), fast_math=self._fast_math())
        self.cc_data.device_modified()
        if self.particles is not None:   # constant velocity field, advection/simulation.py:82-90
            self.advance_particles(g.scratch_array() + self.rp.get_param("advection.u"),
                                   g.scratch_array() + self.rp.get_param("advection.v"))
        self.cc_data.t += self.dt
        self.n += 1
        tm.end()

    def can_evolve_many(self):
        """may the driver hand several steps at once to the device (pyrohip_adv_evolve)?
        Standard boundary types, nothing watching the data, the plain evolve() of this class
        (tracer particles ride along: the velocity field is constant, they never read the
        data)."""
        cc = self.cc_data
        if type(self).evolve is not Simulation.evolve or cc.slab is not None:
            return False      # (a slab exchanges halo rows before every step: fill_BC_all)
        simple = ("outflow", "reflect-even", "reflect-odd", "periodic")
        if not all(b in simple for n in cc.names for b in cc.BCs[n].sides()):
            return False
        return not (any(cc._has_host_bc(n) for n in cc.names) or cc._views_alive())

    def evolve_many(self, nsteps):
        """up to nsteps iterations of fill_BC_all + compute_timestep + evolve
        (pyro_sim.py:250-256) in one device call.  The advective CFL step is closed-form
        (advection/simulation.py:38-54), so the driver's policy (simulation_null.py:222-244)
        gives the whole dt sequence beforehand -- computed here by the very methods the
        single step uses; on periodic grids the device then takes several steps per pass
        over the grid (csrc/advection.hip: k_adv_multi).  Returns the time steps taken."""
        tm = self.tc.timer("evolve")
        tm.begin()
        t0, n0 = self.cc_data.t, self.n
        dts = []
        while len(dts) < nsteps and not self.finished():
            keep = (getattr(self, "dt", None), getattr(self, "dt_old", None))
            self.compute_timestep()
            if not self.dt > 0.0:
                # no positive step to hand to the device: undo this policy call (t, n were not
                # advanced for it) and let the driver take the step singly, like the reference
                self.dt, self.dt_old = keep
                break
            dts.append(float(self.dt))
            self.cc_data.t += self.dt       # as evolve() does
            self.n += 1
        if dts:
            self.cc_data.t, self.n = t0, n0
            g = self.cc_data.grid
            st = self.cc_data.device_state(fuse_fill=True)
            self.cc_data.take_pending_fill()     # every step of the call fills
            try:
                st.adv_evolve(self.cc_data.names.index("density"), g.dx, g.dy,
                              float(self.rp.get_param("advection.u")),
                              float(self.rp.get_param("advection.v")), dts,
                              int(self.rp.get_param("advection.limiter")),
                              fast_math=self._fast_math(), multi_k=self._multi_k())
            finally:
                self.cc_data.device_modified()
            if self.particles is not None:   # advection/simulation.py:82-90, step by step
                uu = g.scratch_array() + self.rp.get_param("advection.u")
                vv = g.scratch_array() + self.rp.get_param("advection.v")
            for dt in dts:                  # the same additions in the same order
                if self.particles is not None:
                    self.dt = dt
                    self.advance_particles(uu, vv)
                self.cc_data.t += dt
            self.n = n0 + len(dts)
        tm.end()
        return dts

    def _multi_k(self):
        """gpu.adv_steps_per_launch (0: the library's choice)"""
        try:
            return int(self.rp.get_param("gpu.adv_steps_per_launch"))
        except (KeyError, ValueError, RuntimeError):
            return 0

    def dovis(self):
        """runtime plot of the density (same picture as the reference's dovis)"""
        import matplotlib.pyplot as plt
        import numpy as np
        plt.clf()
        dens = self.cc_data.get_var("density")
        g = self.cc_data.grid
        img = plt.imshow(np.transpose(dens.v()), interpolation="nearest", origin="lower",
                         extent=[g.xmin, g.xmax, g.ymin, g.ymax], cmap=self.cm)
        plt.xlabel("x")
        plt.ylabel("y")
        plt.colorbar(img)
        plt.title("density")
        plt.figtext(0.05, 0.0125, f"t = {self.cc_data.t:10.5f}")
        plt.pause(0.001)
        plt.draw()
