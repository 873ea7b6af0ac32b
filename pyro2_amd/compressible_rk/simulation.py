"""compressible_rk.Simulation with the call surface of
pyro/compressible_rk/simulation.py:8-104: the compressible solver's state,
problems and boundaries with a method-of-lines update.  Per stage: ghost fill,
pyrohip_comp_rk_rhs (density floor, primitives, flattening, limited slopes,
face states, Riemann fluxes, artificial viscosity, flux divergence, gravity and
sponge sources); stage starts and the final update are
pyrohip_state_lincomb launches (pyro2_amd/mesh/integration.py)."""
from ..compressible.simulation import Simulation as CompressibleSimulation
from ..mesh import integration
from ..util import msg


class Simulation(CompressibleSimulation):
    spherical_ok = False   # compressible_rk/fluxes.py has no geometry terms
    decomposable = False   # (the stages would each need a halo exchange: single domain)

    def initialize(self, *, extra_vars=None, ng=4):
        if self._rp_opt("compressible.well_balanced", 0):
            msg.fail("ERROR: compressible.well_balanced is not carried by the device path")
        super().initialize(extra_vars=extra_vars, ng=ng)
        self._rk_scratch = None

    def substep(self, st, kstate, slot):
        """k of the device state `st` into slot `slot` of `kstate`"""
        st.comp_rk_rhs(self._params(), kstate, slot)

    def method_compute_timestep(self):
        """cfl * min 1 / ((|u|+c)/dx + (|v|+c)/dy) over the whole array
        (compressible_rk/simulation.py:46-56)"""
        cfl = self.rp.get_param("driver.cfl")
        # the one-call step leaves the minimum of the new state: no ghost cells needed for it
        # (the step then does the only ghost fill of the iteration)
        st = self._device_state(fuse_fill=True)
        if not st.comp_rk_dt_is_cached():
            st = self._device_state()
        self.dt = st.comp_rk_dt(self._params(), float(cfl))

    def _rk_fusable(self, start, method):
        """the whole Runge-Kutta step as nstages launches of the row-marching kernel
        (pyrohip_comp_rk_step): standard boundary types filled by the device, no sponge, no
        heating profile, no host-side source, a grid the kernel pays on (the library decides)"""
        cc = self.cc_data
        if self._host_source() or self._heating() is not None or type(self).substep is not Simulation.substep:
            return False
        if any(cc._has_host_bc(n) for n in cc.names):
            return False
        if self._rk_scratch is None or self._rk_scratch[1].nvar != 4 * len(integration.b[method]):
            rk = integration.RKIntegrator(cc.t, self.dt, method=method)
            self._rk_scratch = rk.set_start(start, None)
        return start.comp_rk_can_fuse(self._params(), self._rk_scratch[1], len(integration.b[method]))

    def evolve(self):
        tm = self.tc.timer("evolve")
        tm.begin()
        cc = self.cc_data
        method = self.rp.get_param("compressible.temporal_method")
        start = self._device_state(fuse_fill=True)
        if not self._rk_fusable(start, method):
            start = self._device_state()     # (stage by stage: the deferred fill is carried out)
        if self._rk_fusable(start, method):
            cc.take_pending_fill()           # (the step fills the state's ghost cells itself)
            start.comp_rk_step(self._params(), self._rk_scratch[1], float(self.dt),
                               integration.a[method], integration.b[method])
            cc.device_modified()
            self.advance_particles()
            cc.t += self.dt
            self.n += 1
            tm.end()
            return
        rk = integration.RKIntegrator(cc.t, self.dt, method=method)
        if self._rk_scratch is not None and self._rk_scratch[1].nvar != 4 * rk.nstages():
            self._rk_scratch = None
        self._rk_scratch = rk.set_start(start, self._rk_scratch)
        h = self._heating()
        if h is not None and not getattr(rk.stage, "_heating_set", False):
            rk.stage.set_heating(h[1])
            rk.stage._heating_set = True
        for s in range(rk.nstages()):
            ytmp = rk.get_stage_start(s)
            if s == 0:
                cc.fill_BC_all()
            else:
                cc._push_user_bc(ytmp)
                ytmp.fill_bc(-1)
            self.substep(ytmp, rk.k, s)
            rk.store_increment(s)
        rk.compute_final_update()
        cc.device_modified()
        self.advance_particles()         # compressible_rk/simulation.py:97-98
        cc.t += self.dt
        self.n += 1
        tm.end()

    def can_evolve_many(self):
        """batches of steps on the device (pyrohip_comp_rk_evolve): where the one-call step runs,
        nothing watches the data, no tracer particles"""
        if self.particles is not None or self.cc_data._views_alive():
            return False
        if self.rp.get_param("sponge.do_sponge") or type(self).evolve is not Simulation.evolve:
            return False
        if getattr(self, "_device_stepping_refused", False):
            return False
        method = self.rp.get_param("compressible.temporal_method")
        return self._rk_fusable(self._device_state(), method)

    def evolve_many(self, nsteps):
        from .._lib import PyroHipError
        from ..decomp import DtPolicy
        rp = self.rp
        pol = DtPolicy(self.tmax, rp.get_param("driver.init_tstep_factor"),
                       rp.get_param("driver.max_dt_change"), rp.get_param("driver.fix_dt"))
        pol.t, pol.n = float(self.cc_data.t), int(self.n)
        pol.dt_old = float(getattr(self, "dt_old", -1.e33))
        method = rp.get_param("compressible.temporal_method")
        tm = self.tc.timer("evolve")
        tm.begin()
        st = self._device_state()
        self.cc_data.take_pending_fill()
        try:
            dts = st.comp_rk_evolve(self._params(), self._rk_scratch[1], integration.a[method],
                                    integration.b[method], float(rp.get_param("driver.cfl")), pol, int(nsteps))
        except PyroHipError as e:
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
