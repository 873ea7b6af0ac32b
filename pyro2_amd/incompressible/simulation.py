"""incompressible.Simulation with the call surface of
pyro/incompressible/simulation.py:14-483.

One step (evolve, :200-372 of the reference) on the device:
  1. inc_mac_rhs   limited slopes, edge states with transverse and grad p terms,
                   MAC velocities, RHS div(U_MAC) of the MAC projection
  2. MG solve      rtol 1e-12 (csrc/multigrid.hip)
  3. inc_advect    MAC correction, upwinded states, advective terms,
                   provisional velocities
  4. ghost fill of u, v; inc_proj_rhs: RHS div(U)/dt, guess = old phi
  5. MG solve
  6. inc_proj_update  phi, velocity correction, grad p update; ghost fill
The reference builds a new MG object for every solve; here one device
hierarchy per set of boundary types is kept and re-initialised.
"""
import numpy as np

from .. import device
from ..burgers.simulation import Simulation as burgers_simulation
from ..mesh import boundary as bnd
from ..mesh import patch
from ..simulation_null import bc_setup, grid_setup
from ..util import msg

_MG_BC = {"periodic": "periodic", "neumann": "neumann", "dirichlet": "dirichlet"}


class Simulation(burgers_simulation):
    def initialize(self, *, other_bc=False, aux_vars=()):
        my_grid = grid_setup(self.rp, ng=4)
        if my_grid.nx != my_grid.ny or 2**int(round(np.log2(my_grid.nx))) != my_grid.nx:
            msg.fail("the multigrid solver needs nx = ny = 2^n")
        my_data = patch.CellCenterData2d(my_grid)
        if other_bc:
            self.define_other_bc()
        bc, bc_xodd, bc_yodd = bc_setup(self.rp)
        my_data.register_var("x-velocity", bc_xodd)
        my_data.register_var("y-velocity", bc_yodd)
        # phi: periodic with the velocity, or Neumann when the velocity is
        # Dirichlet (incompressible/simulation.py:40-48)
        phi_bc = None
        if bc.xlb == "periodic":
            phi_bc = bc
        elif bc.xlb == "dirichlet":
            phi_bc = bnd.BC(xlb="neumann", xrb="neumann", ylb="neumann", yrb="neumann")
        if phi_bc is None:
            msg.fail("incompressible: mesh boundaries must be periodic or dirichlet")
        for name in ("phi-MAC", "phi", "gradp_x", "gradp_y"):
            my_data.register_var(name, phi_bc)
        for k, v in aux_vars:
            my_data.set_aux(keyword=k, value=v)
        my_data.create()
        self.cc_data = my_data
        self.setup_particles(bc)         # incompressible/simulation.py:57-60
        self.in_preevolve = False
        self._mgs = {}
        self.mg_cycles = (0, 0)
        self.problem_func(self.cc_data, self.rp)

    # ---- helpers ---------------------------------------------------------
    def _idx(self):
        n = self.cc_data.names.index
        return (n("x-velocity"), n("y-velocity"), n("phi-MAC"), n("phi"), n("gradp_x"),
                n("gradp_y"))

    def _mg(self, bcs):
        g = self.cc_data.grid
        key = tuple(bcs)
        if key not in self._mgs:
            self._mgs[key] = device.DeviceMG(self.cc_data.ctx, g.nx, xmin=g.xmin, xmax=g.xmax,
                                             ymin=g.ymin, ymax=g.ymax,
                                             bcs=[_MG_BC[b] for b in bcs], alpha=0.0, beta=-1.0,
                                             nsmooth=10, nsmooth_bottom=50)
        return self._mgs[key]

    def _fill_velocity(self):
        self.cc_data.fill_BC("x-velocity")
        self.cc_data.fill_BC("y-velocity")

    # ---- reference surface --------------------------------------------------
    def preevolve(self):
        """initial projection of the velocity field, then one throw-away step
        whose grad p is kept (incompressible/simulation.py:77-143)"""
        self.in_preevolve = True
        cc, g = self.cc_data, self.cc_data.grid
        iu, iv, _, iphi, igx, igy = self._idx()
        self._fill_velocity()
        mg = self._mg(("periodic",) * 4)          # the reference hard-codes periodic here
        st = cc.device_state()
        st.inc_proj_rhs(mg, iu, iv, -1, g.dx, g.dy, 1.0, 0)
        mg.solve(rtol=1.e-10)
        st.inc_proj_update(mg, iu, iv, iphi, igx, igy, g.dx, g.dy, 1.0, 0)
        cc.device_modified()
        self._fill_velocity()
        orig = np.array(cc.data)                   # device -> host copy of the state
        self.method_compute_timestep()
        self.evolve()
        new = np.asarray(cc.data)
        orig[:, :, igx] = new[:, :, igx]
        orig[:, :, igy] = new[:, :, igy]
        cc.data[:, :, :] = orig
        if self.verbose > 0:
            print("done with the pre-evolution")
        self.in_preevolve = False

    def evolve(self, other_update_velocity=False, other_source_term=False):
        """other_source_term: the derived solver's viscosity() enters the edge
        states; other_update_velocity: its do_other_update_velocity replaces the
        provisional update (incompressible/simulation.py:159, :174-176, :305-308)"""
        tm = self.tc.timer("evolve")
        tm.begin()
        cc, g = self.cc_data, self.cc_data.grid
        iu, iv, iphim, iphi, igx, igy = self._idx()
        limiter = self.rp.get_param("incompressible.limiter")
        proj_type = self.rp.get_param("incompressible.proj_type")
        mg = self._mg(cc.BCs["phi"].sides())
        st = cc.device_state()

        if self.verbose > 0:
            print("  making MAC velocities")
        nu = self.viscosity() if other_source_term else 0.0
        st.inc_mac_rhs(mg, iu, iv, igx, igy, g.dx, g.dy, self.dt, limiter, nu)
        if self.verbose > 0:
            print("  MAC projection")
        nc1 = mg.solve(rtol=1.e-12)[0]
        if self.verbose > 0:
            print("  making u, v edge states; provisional update of u, v")
        st.inc_advect(mg, iu, iv, iphim, igx, igy, g.dx, g.dy, self.dt,
                      0 if other_update_velocity else proj_type)
        if other_update_velocity:
            self.do_other_update_velocity(st)
        cc.device_modified()
        self._fill_velocity()

        if self.verbose > 0:
            print("  final projection")
        st = cc.device_state()
        st.inc_proj_rhs(mg, iu, iv, iphi, g.dx, g.dy, self.dt, 1)
        nc2 = mg.solve(rtol=1.e-12)[0]
        st.inc_proj_update(mg, iu, iv, iphi, igx, igy, g.dx, g.dy, self.dt, proj_type)
        cc.device_modified()
        self._fill_velocity()
        self.mg_cycles = (nc1, nc2)
        # incompressible/simulation.py:398-399 asks for a derived "velocity" the
        # solver does not define; the stored x-/y-velocity is what is meant
        self.advance_particles()

        if not self.in_preevolve:
            cc.t += self.dt
            self.n += 1
        tm.end()

    def define_other_bc(self):
        """hook of the reference for derived solvers (incompressible_viscous)"""

    def dovis(self):
        import matplotlib.pyplot as plt
        plt.clf()
        plt.rc("font", size=10)
        u = self.cc_data.get_var("x-velocity")
        v = self.cc_data.get_var("y-velocity")
        g = self.cc_data.grid
        vort = g.scratch_array()
        divU = g.scratch_array()
        vort.v()[:, :] = 0.5 * (v.ip(1) - v.ip(-1)) / g.dx - 0.5 * (u.jp(1) - u.jp(-1)) / g.dy
        divU.v()[:, :] = 0.5 * (u.ip(1) - u.ip(-1)) / g.dx + 0.5 * (v.jp(1) - v.jp(-1)) / g.dy
        _, axes = plt.subplots(nrows=2, ncols=2, num=1, clear=True)
        plt.subplots_adjust(hspace=0.25)
        for ax, f, name in zip(axes.flat, (u, v, vort, divU),
                               ("u", "v", r"$\nabla \times U$", r"$\nabla \cdot U$")):
            img = ax.imshow(np.transpose(f.v()), interpolation="nearest", origin="lower",
                            extent=[g.xmin, g.xmax, g.ymin, g.ymax], cmap=self.cm)
            ax.set_xlabel("x")
            ax.set_ylabel("y")
            ax.set_title(name)
            plt.colorbar(img, ax=ax)
        plt.figtext(0.05, 0.0125, f"t = {self.cc_data.t:10.5f}")
        plt.pause(0.001)
        plt.draw()
