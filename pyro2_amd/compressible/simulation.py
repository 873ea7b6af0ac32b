"""compressible.Simulation with the call surface of
pyro/compressible/simulation.py:12-553.

evolve() = pyrohip_comp_step (interface states, HLLC Riemann problems,
transverse correction, artificial viscosity, conservative update in HIP);
method_compute_timestep() = pyrohip_comp_dt (device min-reduction, 8 bytes
D2H).  Scope of the device path: Cartesian
grid, HLLC or CGF Riemann solver, gamma-law gas, limiter 0/1/2, flattening,
artificial viscosity, gravity, sponge, standard boundary types plus the
hse / ambient user boundaries.
"""
import numpy as np

from .. import device
from .._lib import BC_CODE, PyroHipError
from ..mesh import boundary as bnd
from ..simulation_null import NullSimulation, bc_setup, grid_setup
from ..util import msg
from . import BC, derives, eos


class Variables:
    """integer keys of the conserved / primitive components
    (compressible/simulation.py:12-46)"""

    def __init__(self, myd):
        self.nvar = len(myd.names)
        self.idens = myd.names.index("density")
        self.ixmom = myd.names.index("x-momentum")
        self.iymom = myd.names.index("y-momentum")
        self.iener = myd.names.index("energy")
        self.naux = self.nvar - 4
        self.irhox = 4 if self.naux > 0 else -1
        self.nq = 4 + self.naux
        self.irho, self.iu, self.iv, self.ip = 0, 1, 2, 3
        self.ix = 4 if self.naux > 0 else -1


def cons_to_prim(U, gamma, ivars, myg):
    """host-side conversion for analysis scripts (the device kernels do their
    own); same formulas as compressible/simulation.py:49-80"""
    q = myg.scratch_array(nvar=ivars.nq)
    rho = U[:, :, ivars.idens]
    ok = rho != 0.0
    q[:, :, ivars.irho] = rho
    q[:, :, ivars.iu] = np.divide(U[:, :, ivars.ixmom], rho, out=np.zeros_like(rho), where=ok)
    q[:, :, ivars.iv] = np.divide(U[:, :, ivars.iymom], rho, out=np.zeros_like(rho), where=ok)
    e = np.divide(U[:, :, ivars.iener] - 0.5 * rho * (q[:, :, ivars.iu]**2 + q[:, :, ivars.iv]**2),
                  rho, out=np.zeros_like(rho), where=ok)
    q[:, :, ivars.ip] = eos.pres(gamma, rho, e)
    return q


def prim_to_cons(q, gamma, ivars, myg):
    U = myg.scratch_array(nvar=ivars.nvar)
    U[:, :, ivars.idens] = q[:, :, ivars.irho]
    U[:, :, ivars.ixmom] = q[:, :, ivars.iu] * U[:, :, ivars.idens]
    U[:, :, ivars.iymom] = q[:, :, ivars.iv] * U[:, :, ivars.idens]
    U[:, :, ivars.iener] = eos.rhoe(gamma, q[:, :, ivars.ip]) + \
        0.5 * q[:, :, ivars.irho] * (q[:, :, ivars.iu]**2 + q[:, :, ivars.iv]**2)
    return U


class Simulation(NullSimulation):
    spherical_ok = True   # derived solvers without the geometry terms clear this
    decomposable = True   # x-slabs, one process per GPU (derived solvers with other steps clear this)

    def initialize(self, *, extra_vars=None, ng=4):
        my_grid = grid_setup(self.rp, ng=ng, spherical_ok=type(self).spherical_ok,
                             decomposable=type(self).decomposable)
        my_data = self.data_class(my_grid)
        # a bare RuntimeParameters (the reference's unit tests build one by hand) may
        # not carry the solver's parameters yet: they are only needed from the first step on
        riemann_method = self._rp_opt("compressible.riemann", None)
        if riemann_method is not None and riemann_method not in ("HLLC", "CGF", "HLLC_lm"):
            msg.fail("ERROR: Riemann solver undefined")
        if my_grid.coord_type == 1 and riemann_method not in (None, "CGF"):   # simulation.py:206-208
            msg.fail("ERROR: only the CGF Riemann solver is supported "
                     "with SphericalPolar geometry")
        # compressible/simulation.py:212-214; both have device kernels
        bnd.define_bc("hse", BC.user, is_solid=False, device_code=BC_CODE["hse"])
        bnd.define_bc("ambient", BC.user, is_solid=False, device_code=BC_CODE["ambient"])
        bnd.define_bc("ramp", BC.user, is_solid=False, device_code=BC_CODE["ramp"])
        bc, bc_xodd, bc_yodd = bc_setup(self.rp)
        self.solid = bnd.bc_is_solid(bc)
        # same registration order as compressible/simulation.py:223-226
        my_data.register_var("density", bc)
        my_data.register_var("energy", bc)
        my_data.register_var("x-momentum", bc_xodd)
        my_data.register_var("y-momentum", bc_yodd)
        if extra_vars:
            msg.fail("ERROR: passive scalars are not carried by the device path yet")
        my_data.set_aux("gamma", self.rp.get_param("eos.gamma"))
        my_data.set_aux("grav", self.rp.get_param("compressible.grav"))
        my_data.create()
        # fill_BC_all may be folded into the step kernel (pyrohip_comp_params.fuse_fill:
        # on grids below 2048^2 a step is mostly launches); any other look at the data
        # carries the fill out first (mesh/patch.py fill_BC_all)
        my_data.lazy_fill = True
        self.cc_data = my_data
        self.ivars = Variables(my_data)
        self.cc_data.add_derived(derives.derive_primitives)
        # compressible/simulation.py:243-244 hands the parameter object over as
        # the particle count (TypeError in the reference); the count and the
        # generator of the [particles] section are what is meant
        self.setup_particles(bc)
        self.problem_func(self.cc_data, self.rp)
        if self.verbose > 0:
            print(my_data)

    def _params(self):
        rp, g = self.rp, self.cc_data.grid
        opt = self._rp_opt
        return device.make_comp_params(
            g.dx, g.dy, gamma=rp.get_param("eos.gamma"),
            limiter=rp.get_param("compressible.limiter"),
            use_flattening=rp.get_param("compressible.use_flattening"),
            z0=rp.get_param("compressible.z0"), z1=rp.get_param("compressible.z1"),
            delta=rp.get_param("compressible.delta"), cvisc=rp.get_param("compressible.cvisc"),
            grav=rp.get_param("compressible.grav"),
            small_dens=rp.get_param("compressible.small_dens"),
            fast_math=opt("gpu.fast_math", 1), kernel_set=opt("gpu.kernel_set", -1),
            riemann=rp.get_param("compressible.riemann"),
            solid_xl=self.solid.xl, solid_yl=self.solid.yl,
            avisc_xhi_interior=self._slab().avisc_xhi_interior if self.cc_data.slab is not None else 0,
            sponge=(rp.get_param("sponge.sponge_rho_begin"), rp.get_param("sponge.sponge_rho_full"),
                    rp.get_param("sponge.sponge_timescale"))
            if rp.get_param("sponge.do_sponge") else None,
            heat_rate=self._heating()[0] if self._heating() else 0.0)

    def _heating(self):
        """(e_rate, profile) of the problem source rho * e_rate * profile(x, y)
        (problems heating / plume / convection), evaluated once; None without"""
        fn = getattr(self, "problem_heating", None)
        if fn is None or self._rp_opt("gpu.host_source", 0):
            return None
        if getattr(self, "_heat_cache", None) is None:
            rate, prof = fn(self.cc_data.grid, self.rp)
            self._heat_cache = (float(rate), np.ascontiguousarray(prof, dtype=np.float64))
        return self._heat_cache

    def _slab(self):
        """decomposed run: the driver of this rank's x-slab (decomp.SlabCompressible around the
        device state of cc_data: halo exchange before the ghost fill, global CFL minimum,
        boundary strips first + overlapped exchange, device-side stepping); None otherwise"""
        cc = self.cc_data
        if cc.slab is None:
            return None
        if cc._slab_driver is None:
            from ..decomp import SlabCompressible
            sides = list(cc.BCs["density"].sides())
            # (the device state is created, not touched: the boundary table does not depend on
            # the data)
            st = cc.device_state(fuse_fill=True)
            cc._slab_driver = SlabCompressible(cc.ctx, cc.slab, cc.grid.ny, sides, None, cc.comm,
                                               ng=cc.grid.ng, state=st)
        return cc._slab_driver

    def _device_state(self, fuse_fill=False):
        """the state on the device, carrying the heating profile if there is one;
        fuse_fill: a ghost fill that fill_BC_all deferred stays pending (the caller's
        kernel does it, or does not need ghost cells)"""
        st = self.cc_data.device_state(fuse_fill=fuse_fill)
        g = self.cc_data.grid
        if g.coord_type == 1 and getattr(st, "_geometry_set", False) is False:
            st.set_geometry(g.device_geometry(), g.xmin, g.ymin)
            st._geometry_set = True
        h = self._heating()
        if h is not None and getattr(st, "_heating_set", False) is False:
            st.set_heating(h[1])
            st._heating_set = True
        return st

    def method_compute_timestep(self):
        """cfl * min(dx/(|u|+c), dy/(|v|+c)) over the whole array
        (compressible/simulation.py:267-288), reduced on the device"""
        cfl = self.rp.get_param("driver.cfl")
        if self.cc_data.slab is not None:    # COLLECTIVE: the minimum over every rank's slab
            self._device_state()
            self.dt = self._slab().dt(float(cfl), self._params())
            return
        st = self._device_state(fuse_fill=True)
        if not st.comp_dt_is_cached():       # the reduction reads the ghost cells too
            st = self._device_state()
        self.dt = st.comp_dt(self._params(), float(cfl))

    def _host_source(self):
        """is the problem source an arbitrary callback the host has to evaluate
        (no heating_profile companion, or gpu.host_source = 1)?"""
        return self.problem_source is not None and self._heating() is None

    def _source_states(self):
        """two device states for S_h(U^n) and S_h(U*), with the boundary types of the
        reference's aux data (dens_src, E_src: bc; xmom_src: bc_xodd; ymom_src: bc_yodd
        -- simulation.py:248-253 -- i.e. those of the state itself; hse and ambient
        fill the source ghost cells with copies of the last interior row,
        BC.py:56-63,104-110,146-151)"""
        if getattr(self, "_src_states", None) is None:
            cc, g = self.cc_data, self.cc_data.grid
            rows = []
            for name in cc.names:
                sides = list(cc.BCs[name].sides())
                if "ramp" in sides:
                    msg.fail("ERROR: the ramp boundary zeroes the source arrays (BC.py:198-200)")
                rows.append(["outflow" if b in ("hse", "ambient") else b for b in sides])
            self._src_states = tuple(device.DeviceState(cc.ctx, g.nx, g.ny, g.ng, rows)
                                     for _ in range(2))
        return self._src_states

    def _evolve_host_source(self):
        """one step with the problem's source_terms() evaluated on the host, twice, where
        compressible/simulation.py evaluates it (:317-320 through unsplit_fluxes.py:295,
        :406-408 share S(U^n); :416-418 is S(U*)); everything else stays on the device"""
        cc, g, dt, P = self.cc_data, self.cc_data.grid, float(self.dt), self._params()
        if g.coord_type != 0:
            msg.fail("ERROR: host-evaluated source terms need a Cartesian grid")
        src_old, src_new = self._source_states()
        S = self.problem_source(g, cc.data, self.ivars, self.rp)
        src_old.upload(np.ascontiguousarray(S, dtype=np.float64))
        src_old.fill_bc()
        del S
        st = self._device_state()
        st.set_source(0, src_old)
        st.comp_step(P, dt)                              # ... up to U* = U + dt S(U^n)
        cc.device_modified()
        S = self.problem_source(g, cc.data, self.ivars, self.rp)
        src_new.upload(np.ascontiguousarray(S, dtype=np.float64))
        del S
        st = self._device_state()
        st.set_source(1, src_new)
        st.comp_source_correct(P, dt)

    def evolve(self):
        tm = self.tc.timer("evolve")
        tm.begin()
        if self._host_source():
            if self.cc_data.slab is not None:
                msg.fail("ERROR: host-evaluated source terms are not carried by a decomposed run")
            self._device_state()
            self._evolve_host_source()
        else:
            self._slab()
            st = self._device_state(fuse_fill=True)
            P = self._params()
            P.fuse_fill = int(self.cc_data.take_pending_fill())
            st.comp_step(P, float(self.dt))
        self.cc_data.device_modified()
        self.advance_particles()         # compressible/simulation.py:443-444
        self.cc_data.t += self.dt
        self.n += 1
        tm.end()

    def can_evolve_many(self):
        """may the driver hand several steps at once to the device
        (pyrohip_comp_evolve)?  Standard boundary types, no sponge, no tracer particles,
        a fused kernel set, nothing watching the data (a SphericalPolar grid: its one-launch
        step, i.e. no heating profile either)."""
        cc = self.cc_data
        if self.particles is not None or self._host_source():
            return False
        if cc.grid.coord_type != 0 and self._heating() is not None:
            return False
        if self.rp.get_param("sponge.do_sponge") or type(self).evolve is not Simulation.evolve:
            return False
        simple = ("outflow", "reflect-even", "reflect-odd", "periodic")
        if not all(b in simple for n in cc.names for b in cc.BCs[n].sides()):
            return False
        if any(cc._has_host_bc(n) for n in cc.names) or cc._views_alive():
            return False
        if getattr(self, "_device_stepping_refused", False):
            return False
        if cc.slab is not None:
            # the exchange must live in the library (RCCL); hse meeting a periodic cut steps singly
            from ..decomp import RcclComm
            if not isinstance(cc.comm, RcclComm) or self._slab()._hse_wrap:
                return False
        return self._params().kernel_set != 0

    def evolve_many(self, nsteps):
        """up to nsteps of fill_BC_all + compute_timestep + evolve on the device without
        a host round trip per step; the driver's dt policy (simulation_null.py:222-244)
        runs in a kernel.  Returns the time steps taken."""
        from ..decomp import DtPolicy
        rp = self.rp
        pol = DtPolicy(self.tmax, rp.get_param("driver.init_tstep_factor"),
                       rp.get_param("driver.max_dt_change"), rp.get_param("driver.fix_dt"))
        pol.t, pol.n = float(self.cc_data.t), int(self.n)
        pol.dt_old = float(getattr(self, "dt_old", -1.e33))
        tm = self.tc.timer("evolve")
        tm.begin()
        st = self._device_state()
        try:
            if self.cc_data.slab is not None:     # COLLECTIVE
                dts = self._slab().evolve(pol, float(rp.get_param("driver.cfl")), int(nsteps),
                                          params=self._params())
            else:
                dts = st.comp_evolve(self._params(), float(rp.get_param("driver.cfl")), pol, int(nsteps))
        except PyroHipError as e:
            # the library's own fusability rules (e.g. a SphericalPolar grid too small for the
            # tile kernel, mixed boundary kinds on a side) are stricter than can_evolve_many's:
            # a refusal before the first step means "step singly", which the staged set can
            if "device-side stepping:" not in str(e) or pol.n != int(self.n):
                raise
            self._device_stepping_refused = True
            dts = []
        finally:
            self.cc_data.device_modified()
            self.cc_data.t, self.n, self.dt_old = pol.t, pol.n, pol.dt_old
        if len(dts):
            self.dt = float(dts[-1])
        tm.end()
        return dts

    def clean_state(self, U):
        """density floor on a host array (the device step applies it itself)"""
        U.v(n=self.ivars.idens)[:, :] = np.maximum(U.v(n=self.ivars.idens),
                                                   self.rp.get_param("compressible.small_dens"))

    def write_extras(self, f):
        """the BC group of compressible/simulation.py:543-553"""
        gb = f.create_group("BC")
        gb.create_dataset("hse", data=np.array([], dtype=np.float64))

    def dovis(self):
        import matplotlib.pyplot as plt
        plt.clf()
        g = self.cc_data.grid
        rho, u, v, p = self.cc_data.get_var("primitive")
        fields = [rho, np.sqrt(u**2 + v**2), p, p / ((self.rp.get_param("eos.gamma") - 1.0) * rho)]
        names = [r"$\rho$", "U", "p", "e"]
        for n, (f, nm) in enumerate(zip(fields, names)):
            ax = plt.subplot(2, 2, n + 1)
            img = ax.imshow(np.transpose(f.v()), interpolation="nearest", origin="lower",
                            extent=[g.xmin, g.xmax, g.ymin, g.ymax], cmap=self.cm)
            ax.set_title(nm)
            plt.colorbar(img, ax=ax)
        plt.figtext(0.05, 0.0125, f"t = {self.cc_data.t:10.5f}")
        plt.pause(0.001)
        plt.draw()
