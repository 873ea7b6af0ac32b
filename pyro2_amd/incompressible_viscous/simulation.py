"""incompressible_viscous.Simulation with the call surface of
pyro/incompressible_viscous/simulation.py:8-190.

evolve() is the incompressible step with two changes (both on the device):
  * the viscous term nu L(U) enters the edge-state prediction
    (other_source_term, :24-41; csrc/incompressible.hip k_bg_trans)
  * the provisional velocity comes from two Helmholtz solves
    (1 - dt nu / 2 L) w = w + dt nu / 2 L(w) - dt [(U.grad)w (+ grad p)]
    (do_other_update_velocity, :43-176) with the boundary types of u and v.
The reference builds two new MG objects per step; here one hierarchy per set of
boundary types is kept and its beta = dt nu / 2 is reset every step.
"""
from .. import device
from ..incompressible import Simulation as incompressible_simulation
from ..mesh import boundary as bnd
from . import BC

_MG_BC = {"periodic": "periodic", "neumann": "neumann", "dirichlet": "dirichlet",
          "moving_lid": "moving_lid"}


class Simulation(incompressible_simulation):
    def initialize(self):  # pylint: disable=arguments-differ
        nu = self.rp.get_param("incompressible_viscous.viscosity")
        self._visc_mgs = {}
        super().initialize(other_bc=True, aux_vars=(("viscosity", nu),))

    def define_other_bc(self):
        bnd.define_bc("moving_lid", BC.user, is_solid=False,
                      device_code=device.BC_CODE["moving_lid"], const_value=BC.lid_value)

    def evolve(self):  # pylint: disable=arguments-differ
        super().evolve(other_update_velocity=True, other_source_term=True)

    def viscosity(self):
        return self.rp.get_param("incompressible_viscous.viscosity")

    def _visc_mg(self, bcs, beta):
        g = self.cc_data.grid
        key = tuple(bcs)
        if key not in self._visc_mgs:
            self._visc_mgs[key] = device.DeviceMG(
                self.cc_data.ctx, g.nx, xmin=g.xmin, xmax=g.xmax, ymin=g.ymin, ymax=g.ymax,
                bcs=[_MG_BC[b] for b in bcs], alpha=1.0, beta=beta, nsmooth=10,
                nsmooth_bottom=50)
        mg = self._visc_mgs[key]
        mg.set_helmholtz(1.0, beta)
        return mg

    def do_other_update_velocity(self, st):
        """the two parabolic solves; st: the device state after inc_advect with
        proj_type 0 (advective terms in the work area, velocities untouched)"""
        if self.verbose > 0:
            print("  doing parabolic solve for u, v")
        cc, g = self.cc_data, self.cc_data.grid
        iu, iv, _, _, igx, igy = self._idx()
        nu = self.viscosity()
        proj_type = self.rp.get_param("incompressible.proj_type")
        cycles = []
        for comp, (iw, igp, name) in enumerate(((iu, igx, "x-velocity"),
                                                (iv, igy, "y-velocity"))):
            mg = self._visc_mg(cc.BCs[name].sides(), 0.5 * self.dt * nu)
            st.inc_visc_rhs(mg, iw, comp, igp, g.dx, g.dy, self.dt, nu, proj_type)
            cycles.append(mg.solve(rtol=1.e-12)[0])
            st.inc_visc_store(mg, iw)
        self.visc_cycles = tuple(cycles)

    def write_extras(self, f):
        """note the custom BC (the value is "is_solid"), :178-186"""
        gb = f.create_group("BC")
        gb.create_dataset("moving_lid", data=False)
