"""Thin object layer over the C ABI (include/pyrohip.h): handles + NumPy I/O.

These classes own opaque library handles; all arithmetic happens in the HIP
kernels.  The pyro-facing API (Grid2d / CellCenterData2d / Simulation /
CellCenterMG2d mirrors) is built on top of them in pyro2_amd.mesh,
pyro2_amd.advection, pyro2_amd.compressible and pyro2_amd.multigrid.
"""
import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import BC_CODE, CompParams, check, dptr, iptr


def comp_wave_geometry(nx, ny, ng=4, num_cus=0, march_rows=0):
    """launch geometry of the row-marching compressible kernel on an nx x ny grid / slab
    (pyrohip_comp_wave_geometry): dict of column strips, rows per strip, row strips, overlap
    eligibility, wavefronts per launch, resident wavefront slots"""
    out = (C.c_int * 6)()
    check(_lib.lib().pyrohip_comp_wave_geometry(int(nx), int(ny), int(ng), int(num_cus),
                                                  int(march_rows), out))
    return dict(zip(("col_strips", "rows_per_strip", "row_strips", "overlap", "wavefronts", "slots"),
                    (int(v) for v in out)))


def device_count():
    n = C.c_int()
    check(_lib.lib().pyrohip_device_count(C.byref(n)))
    return n.value


class Context:
    """One HIP device + stream.  Calls are serialised with a lock because
    ctypes releases the GIL (include/pyrohip.h conventions)."""

    _default = None

    def __init__(self, device_id=0):
        self._l = _lib.lib()
        self.h = C.c_void_p()
        check(self._l.pyrohip_init(int(device_id), C.byref(self.h)))
        self.device_id = int(device_id)
        self.lock = threading.RLock()

    @classmethod
    def default(cls, device_id=None):
        import os
        if cls._default is None:
            if device_id is None:
                device_id = int(os.environ.get("PYRO2_AMD_DEVICE",
                                               os.environ.get("LOCAL_RANK", "0")))
            cls._default = cls(device_id)
        return cls._default

    def close(self):
        if self.h:
            self._l.pyrohip_shutdown(self.h)
            self.h = C.c_void_p()
            if Context._default is self:
                Context._default = None

    def sync(self):
        with self.lock:
            check(self._l.pyrohip_sync(self.h))

    def info(self):
        name = C.create_string_buffer(256)
        fr, tot, cus = C.c_size_t(), C.c_size_t(), C.c_int()
        with self.lock:
            check(self._l.pyrohip_device_info(self.h, name, 256, C.byref(fr),
                                              C.byref(tot), C.byref(cus)))
        return {"name": name.value.decode(), "free": fr.value,
                "total": tot.value, "compute_units": cus.value,
                "backend": _lib.backend_name()}

    def timer_start(self):
        with self.lock:
            check(self._l.pyrohip_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_double()
        with self.lock:
            check(self._l.pyrohip_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def prof_enable(self, on=True):
        with self.lock:
            check(self._l.pyrohip_prof_enable(self.h, 1 if on else 0))

    def prof_report(self):
        """{kernel: (launches, total_ms)} since the last report"""
        buf = C.create_string_buffer(1 << 16)
        with self.lock:
            check(self._l.pyrohip_prof_report(self.h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split()
            out[name] = (int(n), float(ms))
        return out

    # ---- multi-GPU plumbing (RCCL) ------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        check(_lib.lib().pyrohip_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        with self.lock:
            check(self._l.pyrohip_comm_init(self.h, nranks, rank, unique_id))

    def comm_destroy(self):
        """drop the RCCL communicator(s) of this context (a no-op without one)"""
        with self.lock:
            check(self._l.pyrohip_comm_destroy(self.h))

    def comm_size(self):
        """rank count of the communicator as RCCL reports it (0: none)"""
        n = C.c_int()
        with self.lock:
            check(self._l.pyrohip_comm_size(self.h, C.byref(n)))
        return n.value

    def comm_set_global_dt(self, on=True):
        """let comp_step all-reduce the next CFL minimum on the device"""
        with self.lock:
            check(self._l.pyrohip_comm_set_global_dt(self.h, int(bool(on))))

    def allreduce_min(self, x):
        v = C.c_double(x)
        with self.lock:
            check(self._l.pyrohip_allreduce_min(self.h, C.byref(v)))
        return v.value

    def allreduce_sum(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        with self.lock:
            check(self._l.pyrohip_allreduce_sum(self.h, dptr(v), int(v.size)))
        return v

    def allreduce_max(self, x):
        v = C.c_double(x)
        with self.lock:
            check(self._l.pyrohip_allreduce_max(self.h, C.byref(v)))
        return v.value


def bc_table(bcs_per_var):
    """[[xl,xr,yl,yr] per variable] of names or codes -> int32 (nvar,4)"""
    out = np.zeros((len(bcs_per_var), 4), dtype=np.int32)
    for n, row in enumerate(bcs_per_var):
        for s, b in enumerate(row):
            out[n, s] = BC_CODE[b] if isinstance(b, str) else int(b)
    return out


class DeviceState:
    """planar SoA copy of a CellCenterData2d.data array on the device"""

    def __init__(self, ctx, nx, ny, ng, bcs_per_var):
        self.ctx = ctx
        self._l = ctx._l
        self.nx, self.ny, self.ng = int(nx), int(ny), int(ng)
        self.qx, self.qy = self.nx + 2 * self.ng, self.ny + 2 * self.ng
        self.bc = np.ascontiguousarray(bc_table(bcs_per_var))
        self.nvar = self.bc.shape[0]
        self.h = C.c_void_p()
        with ctx.lock:
            check(self._l.pyrohip_state_create(ctx.h, self.nx, self.ny, self.ng,
                                               self.nvar, iptr(self.bc.reshape(-1)),
                                               C.byref(self.h)))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self._l.pyrohip_state_destroy(self.h)
        except Exception:
            pass

    def _aos(self, a, rows=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        want = (self.qx if rows is None else rows, self.qy, self.nvar)
        if self.nvar == 1 and a.ndim == 2:
            a = a.reshape(a.shape + (1,))
        assert a.shape == want, (a.shape, want)
        return a

    def upload(self, data):
        a = self._aos(data)
        with self.ctx.lock:
            check(self._l.pyrohip_state_upload(self.h, dptr(a)))

    def download(self, out=None):
        if out is None:
            out = np.empty((self.qx, self.qy, self.nvar))
        a = out if out.ndim == 3 else out.reshape(out.shape + (1,))
        with self.ctx.lock:
            check(self._l.pyrohip_state_download(self.h, dptr(a)))
        return out

    def upload_rows(self, i0, data):
        a = self._aos(data, rows=data.shape[0])
        with self.ctx.lock:
            check(self._l.pyrohip_state_upload_rows(self.h, int(i0), a.shape[0], dptr(a)))

    def download_rows(self, i0, ni):
        out = np.empty((ni, self.qy, self.nvar))
        with self.ctx.lock:
            check(self._l.pyrohip_state_download_rows(self.h, int(i0), int(ni), dptr(out)))
        return out

    def upload_var(self, n, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == (self.qx, self.qy)
        with self.ctx.lock:
            check(self._l.pyrohip_state_upload_var(self.h, int(n), dptr(a)))

    def download_var(self, n):
        out = np.empty((self.qx, self.qy))
        with self.ctx.lock:
            check(self._l.pyrohip_state_download_var(self.h, int(n), dptr(out)))
        return out

    def fill_bc(self, n=-1):
        with self.ctx.lock:
            check(self._l.pyrohip_fill_bc(self.h, int(n)))

    # ---- compressible_rk -----------------------------------------------------
    def comp_rk_rhs(self, params, kstate, slot):
        with self.ctx.lock:
            check(self._l.pyrohip_comp_rk_rhs(self.h, C.byref(params), kstate.h, int(slot)))

    def comp_rk_dt(self, params, cfl):
        out = C.c_double()
        with self.ctx.lock:
            check(self._l.pyrohip_comp_rk_dt(self.h, C.byref(params), float(cfl), C.byref(out)))
        return out.value

    def lincomb(self, src, kstate, coefs):
        """self <- src (whole array); interior += coefs[s] * k_s in order"""
        c = np.ascontiguousarray(coefs, dtype=np.float64)
        with self.ctx.lock:
            check(self._l.pyrohip_state_lincomb(self.h, src.h, kstate.h, dptr(c) if len(c) else
                                                dptr(np.zeros(1)), len(c)))

    # ---- shallow water (csrc/swe.hip) ------------------------------------------
    SWE_RIEMANN = {"Roe": 0, "HLLC": 1}

    def swe_dt(self, dx, dy, grav, cfl):
        out = C.c_double()
        with self.ctx.lock:
            check(self._l.pyrohip_swe_dt(self.h, float(dx), float(dy), float(grav), float(cfl),
                                         C.byref(out)))
        return out.value

    def swe_step(self, dx, dy, grav, limiter, riemann, dt, kernel_set=-1, fast_math=0):
        """kernel_set: 0 staged kernels (swe_stage() dumps), 1 one launch per step (default);
        fast_math (one-launch kernel): 1 the contracted build, 0 bit-faithful"""
        r = self.SWE_RIEMANN[riemann] if isinstance(riemann, str) else int(riemann)
        with self.ctx.lock:
            check(self._l.pyrohip_swe_step_ex(self.h, float(dx), float(dy), float(grav), int(limiter),
                                              r, float(dt), int(kernel_set), int(fast_math)))

    def swe_evolve(self, dx, dy, grav, limiter, riemann, cfl, policy, max_steps, fast_math=0):
        """up to max_steps swe steps with the driver's dt policy on the device (as comp_evolve);
        returns the dt of the steps taken"""
        from ._lib import DtPolicyC
        r = self.SWE_RIEMANN[riemann] if isinstance(riemann, str) else int(riemann)
        pc = DtPolicyC(policy.tmax, policy.f0, policy.mx, policy.fix, policy.t, policy.dt_old,
                       policy.n)
        done = C.c_int()
        dts = np.empty(int(max_steps))
        with self.ctx.lock:
            rc = self._l.pyrohip_swe_evolve(self.h, float(dx), float(dy), float(grav), int(limiter), r,
                                            int(fast_math), float(cfl), C.byref(pc), int(max_steps),
                                            C.byref(done), dptr(dts))
        policy.t, policy.dt_old, policy.n = pc.t, pc.dt_old, int(pc.n)
        check(rc)
        return dts[:done.value]

    def swe_stage(self, name):
        names = ("Uxl0", "Uxr0", "Uyl0", "Uyr0", "FxT", "FyT", "Fx", "Fy")
        out = np.zeros((self.qx, self.qy, 4))
        with self.ctx.lock:
            check(self._l.pyrohip_swe_stage_dump(self.h, names.index(name), dptr(out)))
        return out

    # ---- burgers / incompressible (csrc/incompressible.hip) ----------------
    def bg_step(self, iu, iv, dx, dy, dt, limiter):
        with self.ctx.lock:
            check(self._l.pyrohip_bg_step(self.h, int(iu), int(iv), float(dx), float(dy),
                                          float(dt), int(limiter)))

    def inc_mac_rhs(self, mg, iu, iv, igpx, igpy, dx, dy, dt, limiter, nu=0.0):
        out = C.c_double()
        with self.ctx.lock:
            check(self._l.pyrohip_inc_mac_rhs(self.h, mg.h, int(iu), int(iv), int(igpx), int(igpy),
                                              float(dx), float(dy), float(dt), int(limiter),
                                              float(nu), C.byref(out)))
        return out.value

    def inc_visc_rhs(self, mg, iw, comp, igp, dx, dy, dt, nu, proj_type):
        """RHS + guess of the Helmholtz solve of velocity component comp
        (incompressible_viscous do_other_update_velocity); returns ||f||"""
        out = C.c_double()
        with self.ctx.lock:
            check(self._l.pyrohip_inc_visc_rhs(self.h, mg.h, int(iw), int(comp), int(igp),
                                               float(dx), float(dy), float(dt), float(nu),
                                               int(proj_type), C.byref(out)))
        return out.value

    def bgv_predict(self, iu, iv, dx, dy, dt, limiter, eps):
        """burgers_viscous: diffusion-corrected edge states + MAC velocities"""
        with self.ctx.lock:
            check(self._l.pyrohip_bgv_predict(self.h, int(iu), int(iv), float(dx), float(dy),
                                              float(dt), int(limiter), float(eps)))

    def bgv_rhs(self, mg, iw, comp, dx, dy, dt, eps):
        """burgers_viscous: RHS of the Helmholtz solve of component comp; returns ||f||"""
        out = C.c_double()
        with self.ctx.lock:
            check(self._l.pyrohip_bgv_rhs(self.h, mg.h, int(iw), int(comp), float(dx), float(dy),
                                          float(dt), float(eps), C.byref(out)))
        return out.value

    def inc_visc_store(self, mg, iw):
        with self.ctx.lock:
            check(self._l.pyrohip_inc_visc_store(self.h, mg.h, int(iw)))

    def set_geometry(self, arrays, xmin, ymin):
        """SphericalPolar geometry for the compressible solver: arrays = dict of
        the grid's Lx, Ly, Ax, Ay, V, dlogAx, dlogAy, x2d (qx, qy) and the sines
        sint, sinb, sinc (qy) of artificial_viscosity; None removes it"""
        from ._lib import GeomArrays
        with self.ctx.lock:
            if arrays is None:
                check(self._l.pyrohip_state_set_geometry(self.h, None))
                return
            G = GeomArrays()
            keep = []
            for n in GeomArrays.NAMES:
                a = np.ascontiguousarray(arrays[n], dtype=np.float64)
                want = (self.qy,) if n.startswith("sin") else (self.qx, self.qy)
                assert a.shape == want, (n, a.shape, want)
                keep.append(a)
                setattr(G, n, dptr(a))
            G.xmin, G.ymin = float(xmin), float(ymin)
            # the 1-d factors of the arrays (SphericalPolar.device_geometry): the one-launch
            # kernel rebuilds the geometry from them instead of reading the planes
            if "rowf" in arrays and "colf" in arrays:
                rowf = np.ascontiguousarray(arrays["rowf"], dtype=np.float64)
                colf = np.ascontiguousarray(arrays["colf"], dtype=np.float64)
                assert rowf.shape == (7, self.qx) and colf.shape == (4, self.qy), (rowf.shape, colf.shape)
                keep += [rowf, colf]
                G.rowf, G.colf = dptr(rowf), dptr(colf)
            check(self._l.pyrohip_state_set_geometry(self.h, C.byref(G)))

    def set_const_bc(self, n, value):
        """ghost value of variable n on its PYROHIP_BC_CONST ("moving_lid") side"""
        with self.ctx.lock:
            check(self._l.pyrohip_state_set_const_bc(self.h, int(n), float(value)))

    def inc_advect(self, mg, iu, iv, iphimac, igpx, igpy, dx, dy, dt, proj_type):
        with self.ctx.lock:
            check(self._l.pyrohip_inc_advect(self.h, mg.h, int(iu), int(iv), int(iphimac),
                                             int(igpx), int(igpy), float(dx), float(dy), float(dt),
                                             int(proj_type)))

    def inc_proj_rhs(self, mg, iu, iv, iphi, dx, dy, dt, divide_by_dt):
        out = C.c_double()
        with self.ctx.lock:
            check(self._l.pyrohip_inc_proj_rhs(self.h, mg.h, int(iu), int(iv), int(iphi),
                                               float(dx), float(dy), float(dt), int(divide_by_dt),
                                               C.byref(out)))
        return out.value

    def inc_proj_update(self, mg, iu, iv, iphi, igpx, igpy, dx, dy, fac, gp_mode):
        with self.ctx.lock:
            check(self._l.pyrohip_inc_proj_update(self.h, mg.h, int(iu), int(iv), int(iphi),
                                                  int(igpx), int(igpy), float(dx), float(dy),
                                                  float(fac), int(gp_mode)))

    def inc_stage(self, which):
        names = ("u_xl", "u_xr", "u_yl", "u_yr", "v_xl", "v_xr", "v_yl", "v_yr",
                 "u_MAC", "v_MAC", "advect_x", "advect_y")
        k = names.index(which) if isinstance(which, str) else int(which)
        out = np.zeros((self.qx, self.qy))
        with self.ctx.lock:
            check(self._l.pyrohip_inc_stage_dump(self.h, k, dptr(out)))
        return out

    def set_heating(self, profile):
        """heating profile (qx, qy) of the problem source S[E] += rho e_rate profile
        (None removes it); e_rate travels in the comp params (heat_rate)"""
        with self.ctx.lock:
            if profile is None:
                check(self._l.pyrohip_state_set_heating(self.h, None))
            else:
                a = np.ascontiguousarray(profile, dtype=np.float64)
                assert a.shape == (self.qx, self.qy)
                check(self._l.pyrohip_state_set_heating(self.h, dptr(a)))

    def set_ramp_bc(self, x, cxoff, post, pre, sf_down, sf_up):
        """parameters of the "ramp" boundary (compressible/BC.py ramp_params)"""
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, post, pre, sf_down, sf_up)]
        assert arrs[0].size == self.qx and arrs[3].size >= self.ng and arrs[4].size >= self.ng
        with self.ctx.lock:
            check(self._l.pyrohip_state_set_ramp_bc(self.h, dptr(arrs[0]), float(cxoff),
                                                    dptr(arrs[1]), dptr(arrs[2]), dptr(arrs[3]),
                                                    dptr(arrs[4])))

    def set_user_bc(self, gamma, grav, dy, ambient=None):
        """parameters of the hse / ambient boundaries (compressible/BC.py);
        ambient = (rho, u, v, p)"""
        amb = None if ambient is None else np.ascontiguousarray(ambient, dtype=np.float64)
        with self.ctx.lock:
            check(self._l.pyrohip_state_set_user_bc(
                self.h, float(gamma), float(grav), float(dy),
                None if amb is None else dptr(amb)))

    def minmax(self, n, buf=0):
        mn, mx = C.c_double(), C.c_double()
        with self.ctx.lock:
            check(self._l.pyrohip_state_minmax(self.h, int(n), int(buf),
                                               C.byref(mn), C.byref(mx)))
        return mn.value, mx.value

    # ---- solvers ------------------------------------------------------
    def adv_step(self, n, dx, dy, u, v, dt, limiter, fill=False, fast_math=0, march_rows=0):
        """fill: fold the ghost fill of variable n into the step (one launch);
        fast_math: the contracted build (1e-12) instead of the bit-faithful one"""
        from ._lib import AdvParams
        ap = AdvParams(dx, dy, u, v, int(limiter), int(bool(fill)), int(fast_math), int(march_rows))
        with self.ctx.lock:
            check(self._l.pyrohip_adv_step_p(self.h, int(n), C.byref(ap), dt))

    def adv_evolve(self, n, dx, dy, u, v, dts, limiter, fast_math=0, march_rows=0, multi_k=0,
                   multi_prio=0):
        """len(dts) x (ghost fill + step) of variable n without a host round trip; on
        periodic grids several steps per pass over the grid (pyrohip_adv_evolve)"""
        from ._lib import AdvParams
        ap = AdvParams(dx, dy, u, v, int(limiter), 1, int(fast_math), int(march_rows), int(multi_k),
                       int(multi_prio))
        arr = (C.c_double * len(dts))(*[float(d) for d in dts])
        with self.ctx.lock:
            check(self._l.pyrohip_adv_evolve(self.h, int(n), C.byref(ap), arr, len(dts)))

    def comp_dt(self, params, cfl):
        dt = C.c_double()
        with self.ctx.lock:
            check(self._l.pyrohip_comp_dt(self.h, C.byref(params), cfl, C.byref(dt)))
        return dt.value

    def comp_dt_is_cached(self):
        """will comp_dt answer from the CFL minimum of the last step (no look at the state)?"""
        f = C.c_int()
        check(self._l.pyrohip_comp_dt_is_cached(self.h, C.byref(f)))
        return bool(f.value)

    def comp_rk_dt_is_cached(self):
        """will comp_rk_dt answer from the minimum the last one-call Runge-Kutta step left
        (no look at the state or its ghost cells)?"""
        f = C.c_int()
        with self.ctx.lock:
            check(self._l.pyrohip_comp_rk_dt_is_cached(self.h, C.byref(f)))
        return bool(f.value)

    def comp_dt_is_global(self):
        f = C.c_int()
        check(self._l.pyrohip_comp_dt_is_global(self.h, C.byref(f)))
        return bool(f.value)

    def comp_step(self, params, dt):
        with self.ctx.lock:
            check(self._l.pyrohip_comp_step(self.h, C.byref(params), dt))

    def set_source(self, which, src):
        """host-evaluated problem source: `src` is a 4-variable DeviceState holding
        S_h(U^n), ghost-filled (which = 0) or S_h(U*) (which = 1); None removes it"""
        with self.ctx.lock:
            check(self._l.pyrohip_state_set_source(self.h, int(which), src.h if src is not None else None))
        self._src = getattr(self, "_src", {})
        self._src[which] = src          # borrowed by the library: keep it alive

    def comp_source_correct(self, params, dt):
        with self.ctx.lock:
            check(self._l.pyrohip_comp_source_correct(self.h, C.byref(params), dt))

    def comp_rk_can_fuse(self, params, kstate, nstages):
        """may the whole Runge-Kutta step run as nstages launches (pyrohip_comp_rk_step)?"""
        f = C.c_int()
        with self.ctx.lock:
            check(self._l.pyrohip_comp_rk_can_fuse(self.h, C.byref(params), kstate.h, int(nstages), C.byref(f)))
        return bool(f.value)

    def comp_rk_step(self, params, kstate, dt, a, b):
        """one compressible_rk step: Butcher tableau a (n x n), b (n); kstate: scratch with 4 n planes"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        with self.ctx.lock:
            check(self._l.pyrohip_comp_rk_step(self.h, C.byref(params), kstate.h, float(dt), len(b),
                                               dptr(a), dptr(b)))

    def comp_rk_evolve(self, params, kstate, a, b, cfl, policy, max_steps):
        """up to max_steps compressible_rk steps with the driver's dt policy on the device
        (as comp_evolve); returns the dt of the steps taken"""
        from ._lib import DtPolicyC
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        pc = DtPolicyC(policy.tmax, policy.f0, policy.mx, policy.fix, policy.t, policy.dt_old,
                       policy.n)
        done = C.c_int()
        dts = np.empty(int(max_steps))
        with self.ctx.lock:
            rc = self._l.pyrohip_comp_rk_evolve(self.h, C.byref(params), kstate.h, len(b), dptr(a), dptr(b),
                                                float(cfl), C.byref(pc), int(max_steps), C.byref(done),
                                                dptr(dts))
        policy.t, policy.dt_old, policy.n = pc.t, pc.dt_old, int(pc.n)
        check(rc)
        return dts[:done.value]

    def comp_evolve(self, params, cfl, policy, max_steps):
        """up to max_steps single_steps (ghost fill, dt policy, evolve) without a host
        round trip per step.  `policy`: an object with tmax, f0 (init_tstep_factor), mx
        (max_dt_change), fix, t, dt_old, n (decomp.DtPolicy / helpers.DtPolicy); it is
        advanced in place.  Returns the dt of the steps taken."""
        from ._lib import DtPolicyC
        pc = DtPolicyC(policy.tmax, policy.f0, policy.mx, policy.fix, policy.t, policy.dt_old,
                       policy.n)
        done = C.c_int()
        dts = np.empty(int(max_steps))
        with self.ctx.lock:
            rc = self._l.pyrohip_comp_evolve(self.h, C.byref(params), float(cfl), C.byref(pc),
                                             int(max_steps), C.byref(done), dptr(dts))
        policy.t, policy.dt_old, policy.n = pc.t, pc.dt_old, int(pc.n)
        check(rc)
        return dts[:done.value]

    STAGES = {"q": 0, "xi": 1, "XM": 2, "XP": 3, "YM": 4, "YP": 5, "FxT": 6,
              "FyT": 7, "Fx": 8, "Fy": 9}

    def comp_stage(self, name):
        sid = self.STAGES[name]
        ncomp = 1 if name == "xi" else 4
        out = np.empty((self.qx, self.qy, ncomp))
        with self.ctx.lock:
            check(self._l.pyrohip_comp_stage_dump(self.h, sid, dptr(out)))
        return out[:, :, 0] if ncomp == 1 else out

    def set_neighbours(self, rank_lo, rank_hi):
        """x neighbours of this slab: enables the overlapped halo exchange"""
        with self.ctx.lock:
            check(self._l.pyrohip_state_set_neighbours(self.h, int(rank_lo), int(rank_hi)))

    def halo_pending(self):
        f = C.c_int()
        check(self._l.pyrohip_state_halo_pending(self.h, C.byref(f)))
        return bool(f.value)

    def halo_exchange(self, rank_lo, rank_hi):
        with self.ctx.lock:
            check(self._l.pyrohip_halo_exchange(self.h, int(rank_lo), int(rank_hi)))

    def send_rows(self, i0, ni, peer):
        """whole rows [i0, i0 + ni) of every variable to a peer rank (RCCL; inside a
        pyrohip_comm_group bracket)"""
        with self.ctx.lock:
            check(self._l.pyrohip_state_send_rows(self.h, int(i0), int(ni), int(peer)))

    def recv_rows(self, i0, ni, peer):
        with self.ctx.lock:
            check(self._l.pyrohip_state_recv_rows(self.h, int(i0), int(ni), int(peer)))


def make_comp_params(dx, dy, gamma=1.4, limiter=2, use_flattening=1, z0=0.75,
                     z1=0.85, delta=0.33, cvisc=0.1, grav=0.0,
                     small_dens=-1.e200, avisc_xhi_interior=0,
                     avisc_yhi_interior=0, fast_math=0, kernel_set=0, riemann="HLLC",
                     solid_xl=0, solid_yl=0, sponge=None, heat_rate=0.0, march_rows=0,
                     fuse_fill=0, step_launches=0):
    p = CompParams()
    p.dx, p.dy, p.gamma = dx, dy, gamma
    p.limiter, p.use_flattening = int(limiter), int(use_flattening)
    p.z0, p.z1, p.delta, p.cvisc = z0, z1, delta, cvisc
    p.grav, p.small_dens = grav, small_dens
    p.avisc_xhi_interior = int(avisc_xhi_interior)
    p.avisc_yhi_interior = int(avisc_yhi_interior)
    p.fast_math, p.kernel_set = int(fast_math), int(kernel_set)
    p.riemann = {"HLLC": 0, "CGF": 1, "HLLC_lm": 2}[riemann] if isinstance(riemann, str) else int(riemann)
    p.solid_xl, p.solid_yl = int(solid_xl), int(solid_yl)
    if sponge is not None:   # (rho_begin, rho_full, timescale)
        p.do_sponge = 1
        p.sponge_rho_begin, p.sponge_rho_full, p.sponge_timescale = sponge
    p.heat_rate = float(heat_rate)     # used when the state carries a heating profile
    p.march_rows = int(march_rows)     # kernel_set 2 only; 0 = automatic
    p.fuse_fill = int(fuse_fill)       # 1: the step applies the boundary rules itself
    p.step_launches = int(step_launches)   # comp_evolve, row-marching kernel: 1 = one launch per step, 3 (= 0) = fill / policy / step
    return p


class DeviceMG:
    """device-resident level hierarchy of MG.CellCenterMG2d"""

    def __init__(self, ctx, nx, xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0,
                 bcs=("dirichlet",) * 4, alpha=0.0, beta=-1.0, nsmooth=10,
                 nsmooth_bottom=50, tuning=None):
        self.ctx = ctx
        self._l = ctx._l
        bc = np.array([BC_CODE[b] if isinstance(b, str) else int(b) for b in bcs],
                      dtype=np.int32)
        self.h = C.c_void_p()
        with ctx.lock:
            check(self._l.pyrohip_mg_create(ctx.h, int(nx), xmin, xmax, ymin, ymax,
                                            iptr(bc), alpha, beta, int(nsmooth),
                                            int(nsmooth_bottom), C.byref(self.h)))
            nl = C.c_int()
            check(self._l.pyrohip_mg_nlevels(self.h, C.byref(nl)))
        self.nx = int(nx)
        self.nlevels = nl.value
        self.dx = (xmax - xmin) / nx
        self.source_norm = 0.0
        if tuning:
            self.set_tuning(**tuning)

    def get_tuning(self):
        """the kernels' tuning values (pyrohip_mg_tuning) as a dict"""
        from ._lib import MGTuning
        t = MGTuning()
        with self.ctx.lock:
            check(self._l.pyrohip_mg_get_tuning(self.h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in MGTuning._fields_}

    def tail_counts(self):
        """marching launches so far that carried (residual + restriction, the solve sums)"""
        a, b = C.c_int(0), C.c_int(0)
        with self.ctx.lock:
            check(self._l.pyrohip_mg_tail_counts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_tuning(self, **kw):
        """change some of the tuning values (tests / developer tools; results do not depend
        on them)"""
        from ._lib import MGTuning
        t = MGTuning()
        with self.ctx.lock:
            check(self._l.pyrohip_mg_get_tuning(self.h, C.byref(t)))
            for k, v in kw.items():
                if not hasattr(t, k):
                    raise KeyError(k)
                setattr(t, k, v)
            check(self._l.pyrohip_mg_set_tuning(self.h, C.byref(t)))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self._l.pyrohip_mg_destroy(self.h)
        except Exception:
            pass

    def _n(self, level):
        return 2 ** (level + 1) + 2

    def set(self, level, var, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == (self._n(level),) * 2
        with self.ctx.lock:
            check(self._l.pyrohip_mg_set(self.h, level, var, dptr(a)))

    def get(self, level, var):
        out = np.empty((self._n(level),) * 2)
        with self.ctx.lock:
            check(self._l.pyrohip_mg_get(self.h, level, var, dptr(out)))
        return out

    def set_bcval(self, side, vals):
        with self.ctx.lock:
            if vals is None:
                check(self._l.pyrohip_mg_set_bcval(self.h, side, None))
            else:
                v = np.ascontiguousarray(vals, dtype=np.float64)
                assert v.shape == (self.nx + 2,)
                check(self._l.pyrohip_mg_set_bcval(self.h, side, dptr(v)))

    def _call(self, fn, *args):
        with self.ctx.lock:
            check(getattr(self._l, fn)(self.h, *args))

    def set_smoother(self, kind):
        self._call("pyrohip_mg_set_smoother", int(kind))

    # ---- row windows (slab-decomposed V-cycle, multigrid/slab.py) ----------
    def smooth_rows(self, level, nsweeps, row0, row1, prolong=False):
        self._call("pyrohip_mg_smooth_rows", int(level), int(nsweeps), int(row0), int(row1),
                   int(bool(prolong)))

    def rows_kmax(self, level):
        """red-black iterations ONE row-window launch does on that level (10, 5 or 0)"""
        k = C.c_int()
        self._call("pyrohip_mg_rows_kmax", int(level), C.byref(k))
        return k.value

    def diag_rows(self, row0, row1):
        """(sum ((v - old) / (v + small))^2, sum r^2) over rows [row0, row1] of the finest
        level; old <- v there"""
        out = np.zeros(2)
        self._call("pyrohip_mg_diag_rows", int(row0), int(row1), dptr(out))
        return float(out[0]), float(out[1])

    def save_old(self):
        self._call("pyrohip_mg_save_old")

    def residual_restrict_rows(self, fine, crow0, crow1):
        self._call("pyrohip_mg_residual_restrict_rows", int(fine), int(crow0), int(crow1))

    def get_rows(self, level, var, i0, ni):
        out = np.empty((int(ni), self._n(level)))
        self._call("pyrohip_mg_get_rows", int(level), int(var), int(i0), int(ni), dptr(out))
        return out

    def set_rows(self, level, var, i0, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.ndim == 2 and a.shape[1] == self._n(level)
        self._call("pyrohip_mg_set_rows", int(level), int(var), int(i0), a.shape[0], dptr(a))

    def mark_zero(self, level):
        self._call("pyrohip_mg_mark_zero", int(level))

    def set_helmholtz(self, alpha, beta):
        """new constant coefficients of (alpha - beta L) phi = f"""
        self._call("pyrohip_mg_set_helmholtz", float(alpha), float(beta))
        self.alpha, self.beta = float(alpha), float(beta)

    def zero(self, level, var):
        self._call("pyrohip_mg_zero", level, var)

    def fill_bc(self, level, var=0):
        self._call("pyrohip_mg_fill_bc", level, var)

    def smooth(self, level, n):
        self._call("pyrohip_mg_smooth", level, n)

    def residual(self, level):
        self._call("pyrohip_mg_residual", level)

    def restrict(self, fine):
        self._call("pyrohip_mg_restrict", fine)

    def prolong_add(self, fine):
        self._call("pyrohip_mg_prolong_add", fine)

    def vcycle(self, level=None):
        self._call("pyrohip_mg_vcycle", self.nlevels - 1 if level is None else level)

    def norm(self, level, var):
        out = C.c_double()
        self._call("pyrohip_mg_norm", level, var, C.byref(out))
        return out.value

    def init_rhs_norm(self):
        out = C.c_double()
        self._call("pyrohip_mg_init_rhs_norm", C.byref(out))
        self.source_norm = out.value
        return out.value

    def set_coeffs(self, coeffs, coeffs_bcs):
        """switch to variable-coefficient mode (VarCoeffCCMG2d)"""
        a = np.ascontiguousarray(coeffs, dtype=np.float64)
        assert a.shape == (self.nx + 2,) * 2
        bc = np.array([BC_CODE[b] if isinstance(b, str) else int(b) for b in coeffs_bcs],
                      dtype=np.int32)
        self._call("pyrohip_mg_set_coeffs", dptr(a), iptr(bc))

    def set_general_coeffs(self, alpha, beta, gamma_x, gamma_y, coeffs_bcs):
        """switch to general mode (GeneralMG2d); coeffs_bcs: 4 BC names per
        coefficient in the order alpha, beta, gamma_x, gamma_y"""
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (alpha, beta, gamma_x, gamma_y)]
        for a in arrs:
            assert a.shape == (self.nx + 2,) * 2
        bc = np.array([BC_CODE[b] if isinstance(b, str) else int(b)
                       for row in coeffs_bcs for b in row], dtype=np.int32)
        assert bc.size == 16
        self._call("pyrohip_mg_set_general_coeffs", *[dptr(a) for a in arrs], iptr(bc))

    def set_rhs_cn(self, state, n, coef):
        """f <- phi + coef * L(phi) from variable n of a device state; returns ||f||"""
        out = C.c_double()
        self._call("pyrohip_mg_set_rhs_cn", state.h, int(n), float(coef), C.byref(out))
        return out.value

    def copy_solution(self, state, n):
        self._call("pyrohip_mg_copy_solution", state.h, int(n))

    def solve(self, rtol=1.e-11, max_cycles=100):
        nc, res, rel = C.c_int(), C.c_double(), C.c_double()
        self._call("pyrohip_mg_solve", rtol, int(max_cycles), C.byref(nc),
                   C.byref(res), C.byref(rel))
        return nc.value, res.value, rel.value
