"""swe.Simulation with the call surface of pyro/swe/simulation.py:14-279.
evolve() = pyrohip_swe_step (primitives, limited slopes, characteristic
tracing, transverse Riemann problems, corrected states, final Riemann problems,
conservative update); method_compute_timestep() = pyrohip_swe_dt."""
import numpy as np

from ..mesh import boundary as bnd
from ..mesh import patch
from ..simulation_null import NullSimulation, bc_setup, grid_setup
from ..util import msg
from . import derives


class Variables:
    """integer keys of the conserved / primitive components
    (swe/simulation.py:14-45)"""

    def __init__(self, myd):
        self.nvar = len(myd.names)
        self.ixmom = myd.names.index("x-momentum")
        self.iymom = myd.names.index("y-momentum")
        self.naux = self.nvar - 3
        self.ihx = 3 if self.naux > 0 else -1
        self.nq = 3 + self.naux
        self.ih, self.iu, self.iv = 0, 1, 2
        self.ix = 3 if self.naux > 0 else -1


def cons_to_prim(U, ivars, myg):
    """host-side conversion for analysis scripts (swe/simulation.py:48-63)"""
    q = myg.scratch_array(nvar=ivars.nq)
    q[:, :, ivars.ih] = U[:, :, ivars.ih]
    q[:, :, ivars.iu] = U[:, :, ivars.ixmom] / U[:, :, ivars.ih]
    q[:, :, ivars.iv] = U[:, :, ivars.iymom] / U[:, :, ivars.ih]
    for k in range(ivars.naux):
        q[:, :, ivars.ix + k] = U[:, :, ivars.ihx + k] / q[:, :, ivars.ih]
    return q


def prim_to_cons(q, ivars, myg):
    """swe/simulation.py:65-80"""
    U = myg.scratch_array(nvar=ivars.nvar)
    U[:, :, ivars.ih] = q[:, :, ivars.ih]
    U[:, :, ivars.ixmom] = q[:, :, ivars.iu] * U[:, :, ivars.ih]
    U[:, :, ivars.iymom] = q[:, :, ivars.iv] * U[:, :, ivars.ih]
    for k in range(ivars.naux):
        U[:, :, ivars.ihx + k] = q[:, :, ivars.ix + k] * q[:, :, ivars.ih]
    return U


class Simulation(NullSimulation):
    def initialize(self, *, extra_vars=None, ng=4):
        my_grid = grid_setup(self.rp, ng=ng)
        my_data = patch.CellCenterData2d(my_grid)
        bc, bc_xodd, bc_yodd = bc_setup(self.rp)
        self.solid = bnd.bc_is_solid(bc)
        # registration order of swe/simulation.py:107-110
        my_data.register_var("height", bc)
        my_data.register_var("x-momentum", bc_xodd)
        my_data.register_var("y-momentum", bc_yodd)
        my_data.register_var("fuel", bc)
        if extra_vars:
            msg.fail("ERROR: additional advected scalars are not carried by the device path")
        if self._rp_opt("swe.use_flattening", 0):
            msg.fail("ERROR: swe.use_flattening needs a pressure variable the swe state "
                     "does not have (it fails in the reference too)")
        if self._rp_opt("swe.riemann", "Roe") not in ("Roe", "HLLC"):
            msg.fail("ERROR: Riemann solver undefined")
        my_data.set_aux("g", self.rp.get_param("swe.grav"))
        my_data.create()
        self.cc_data = my_data
        self.setup_particles(bc)         # swe/simulation.py:129-132
        self.ivars = Variables(my_data)
        self.cc_data.add_derived(derives.derive_primitives)
        self.problem_func(self.cc_data, self.rp)
        if self.verbose > 0:
            print(my_data)

    def method_compute_timestep(self):
        """cfl * min(dx/(|u|+c), dy/(|v|+c)) over the whole array, c = sqrt(g h)
        (swe/simulation.py:143-153)"""
        g = self.cc_data.grid
        self.dt = self.cc_data.device_state().swe_dt(
            g.dx, g.dy, self.rp.get_param("swe.grav"), self.rp.get_param("driver.cfl"))

    def evolve(self):
        tm = self.tc.timer("evolve")
        tm.begin()
        cc, g = self.cc_data, self.cc_data.grid
        cc.device_state().swe_step(g.dx, g.dy, self.rp.get_param("swe.grav"),
                                   self.rp.get_param("swe.limiter"),
                                   self.rp.get_param("swe.riemann"), self.dt,
                                   kernel_set=self._rp_opt("gpu.kernel_set", -1) if
                                   self._rp_opt("gpu.kernel_set", -1) in (0, 1) else -1,
                                   fast_math=self._fast_math())
        cc.device_modified()
        self.advance_particles()         # swe/simulation.py:198-199 (derived "velocity")
        cc.t += self.dt
        self.n += 1
        tm.end()

    def _fast_math(self):
        """gpu.fast_math (default 1: the contracted one-launch kernel, <= 1e-10 element-wise of the
        bit-faithful one; 0: the reference's operation order)"""
        fm = int(self._rp_opt("gpu.fast_math", 1))
        if fm and self._rp_opt("gpu.kernel_set", -1) == 0:
            # the staged kernels (every stage dumpable) exist in the reference's operation order only
            if not getattr(self, "_warned_staged", False):
                self._warned_staged = True
                msg.warning("gpu.kernel_set = 0 (staged kernels) runs the bit-faithful arithmetic: "
                            "gpu.fast_math = 1 is ignored")
            return 0
        return fm

    def can_evolve_many(self):
        """batches of steps on the device (pyrohip_swe_evolve): standard boundary types filled
        by the device, nothing watching the data, no tracer particles, the plain evolve()"""
        cc = self.cc_data
        if self.particles is not None or cc._views_alive() or type(self).evolve is not Simulation.evolve:
            return False
        simple = ("outflow", "reflect-even", "reflect-odd", "periodic")
        if not all(b in simple for n in cc.names for b in cc.BCs[n].sides()):
            return False
        if self._rp_opt("gpu.kernel_set", -1) == 0:
            return False
        return not any(cc._has_host_bc(n) for n in cc.names)

    def evolve_many(self, nsteps):
        from ..decomp import DtPolicy
        rp = self.rp
        pol = DtPolicy(self.tmax, rp.get_param("driver.init_tstep_factor"),
                       rp.get_param("driver.max_dt_change"), rp.get_param("driver.fix_dt"))
        pol.t, pol.n = float(self.cc_data.t), int(self.n)
        pol.dt_old = float(getattr(self, "dt_old", -1.e33))
        tm = self.tc.timer("evolve")
        tm.begin()
        cc, g = self.cc_data, self.cc_data.grid
        st = cc.device_state()
        cc.take_pending_fill()
        try:
            dts = st.swe_evolve(g.dx, g.dy, rp.get_param("swe.grav"), rp.get_param("swe.limiter"),
                                rp.get_param("swe.riemann"), float(rp.get_param("driver.cfl")), pol,
                                int(nsteps), fast_math=self._fast_math())
        finally:
            cc.device_modified()
            cc.t, self.n, self.dt_old = pol.t, pol.n, pol.dt_old
        if len(dts):
            self.dt = float(dts[-1])
        tm.end()
        return dts

    def dovis(self):
        import matplotlib.pyplot as plt
        plt.clf()
        plt.rc("font", size=10)
        g = self.cc_data.grid
        q = cons_to_prim(self.cc_data.data, self.ivars, g)
        h, u, v, fuel = (q[:, :, n] for n in range(4))
        vort = g.scratch_array()
        vort.v()[:, :] = 0.5 * (v.ip(1) - v.ip(-1)) / g.dx - 0.5 * (u.jp(1) - u.jp(-1)) / g.dy
        _, axes = plt.subplots(nrows=2, ncols=2, num=1, clear=True)
        for ax, f, name in zip(axes.flat, (h, np.sqrt(u**2 + v**2), fuel, vort),
                               (r"$h$", r"$|U|$", r"$X$", r"$\nabla\times U$")):
            img = ax.imshow(np.transpose(f.v()), interpolation="nearest", origin="lower",
                            extent=[g.xmin, g.xmax, g.ymin, g.ymax], cmap=self.cm)
            ax.set_xlabel("x")
            ax.set_ylabel("y")
            ax.set_title(name)
            plt.colorbar(img, ax=ax)
        plt.figtext(0.05, 0.0125, f"t = {self.cc_data.t:10.5f}")
        plt.pause(0.001)
        plt.draw()
