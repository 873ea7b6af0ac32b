"""ctypes front-end for oracle/pyro_oracle.c.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg as the checker.  The product path (pyro2_amd/)
never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liborc.so")

BC_CODES = {"outflow": 0, "neumann": 0, "reflect-even": 1, "reflect-odd": 2,
            "dirichlet": 2, "periodic": 3, "hse": 5, "ambient": 6, "ramp": 7, "moving_lid": 8}


def build(force=False):
    src = os.path.join(_HERE, "pyro_oracle.c")
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB


class CompParams(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("ng", C.c_int),
                ("dx", C.c_double), ("dy", C.c_double), ("gamma", C.c_double),
                ("limiter", C.c_int), ("use_flattening", C.c_int),
                ("z0", C.c_double), ("z1", C.c_double), ("delta", C.c_double),
                ("cvisc", C.c_double), ("grav", C.c_double),
                ("small_dens", C.c_double), ("bc", (C.c_int * 4) * 4),
                ("avisc_xhi_interior", C.c_int),
                ("avisc_yhi_interior", C.c_int),
                ("riemann", C.c_int), ("solid_xl", C.c_int), ("solid_xr", C.c_int),
                ("solid_yl", C.c_int), ("solid_yr", C.c_int), ("do_sponge", C.c_int),
                ("sponge_rho_begin", C.c_double), ("sponge_rho_full", C.c_double),
                ("sponge_timescale", C.c_double),
                ("heat_rate", C.c_double), ("heat_prof", C.POINTER(C.c_double)),
                ("geom", C.c_void_p)]


class Geom(C.Structure):
    """orc_geom: the arrays of patch.SphericalPolar"""
    _NAMES = ("Lx", "Ly", "Ax", "Ay", "V", "dlogAx", "dlogAy", "x2d", "sint", "sinb", "sinc")
    _fields_ = [(n, C.POINTER(C.c_double)) for n in _NAMES] + \
               [("xmin", C.c_double), ("ymin", C.c_double)]

    def __init__(self, arrays, xmin, ymin):
        super().__init__()
        self._keep = {}
        for n in self._NAMES:
            a = np.ascontiguousarray(arrays[n], dtype=np.float64)
            self._keep[n] = a
            setattr(self, n, _p(a))
        self.xmin, self.ymin = float(xmin), float(ymin)


_STAGE_NAMES = ["q", "xi", "ldx", "ldy", "Uxl0", "Uxr0", "Uyl0", "Uyr0",
                "FxT", "FyT", "Uxl", "Uxr", "Uyl", "Uyr", "Fx0", "Fy0",
                "avx", "avy", "Fx", "Fy"]
_SCALAR_STAGES = {"xi", "avx", "avy"}


class CompStages(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_double)) for n in _STAGE_NAMES]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_adv_dt.restype = C.c_double
        _lib.orc_comp_dt.restype = C.c_double
        _lib.orc_mg_create.restype = C.c_void_p
        _lib.orc_mg_ptr.restype = C.POINTER(C.c_double)
        _lib.orc_mg_norm.restype = C.c_double
        _lib.orc_mg_get_scalar.restype = C.c_double
        _lib.orc_vcmg_create.restype = C.c_void_p
        _lib.orc_vcmg_base.restype = C.c_void_p
        _lib.orc_vcmg_ptr.restype = C.POINTER(C.c_double)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ck(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a


def bc_codes(bcs):
    """('outflow','outflow','reflect-even','periodic') -> int32[4]"""
    return np.array([BC_CODES[b] if isinstance(b, str) else int(b)
                     for b in bcs], dtype=np.int32)


def fill_ghost(a, nx, ny, ng, bcs, n=0):
    """a: (qx,qy) or (qx,qy,nvar) float64, in place."""
    _ck(a)
    nvar = 1 if a.ndim == 2 else a.shape[2]
    bc = bc_codes(bcs)
    lib().orc_fill_ghost(_p(a), nx, ny, ng, nvar, n,
                         bc.ctypes.data_as(C.POINTER(C.c_int)))
    return a


def limit(a, nx, ny, ng, idir, limiter):
    _ck(a)
    out = np.zeros_like(a)
    lib().orc_limit(_p(a), 1, nx, ny, ng, idir, limiter, _p(out))
    return out


def adv_dt(dx, dy, u, v, cfl):
    return lib().orc_adv_dt(C.c_double(dx), C.c_double(dy), C.c_double(u),
                            C.c_double(v), C.c_double(cfl))


def adv_step(a, nx, ny, ng, dx, dy, u, v, dt, limiter, stages=False):
    """one advection step in place on a (qx,qy) (ghosts must be filled)."""
    _ck(a)
    outs = {}
    ptrs = []
    for name in ("ldx", "ldy", "ax", "ay", "Fx", "Fy"):
        if stages:
            outs[name] = np.zeros_like(a)
            ptrs.append(_p(outs[name]))
        else:
            ptrs.append(None)
    lib().orc_adv_step(_p(a), nx, ny, ng, C.c_double(dx), C.c_double(dy),
                       C.c_double(u), C.c_double(v), C.c_double(dt), limiter,
                       *ptrs)
    return outs


def comp_params(nx, ny, ng, dx, dy, gamma=1.4, limiter=2, use_flattening=1,
                z0=0.75, z1=0.85, delta=0.33, cvisc=0.1, grav=0.0,
                small_dens=-1.e200, bcs=("outflow",) * 4,
                avisc_xhi_interior=0, avisc_yhi_interior=0, riemann="HLLC",
                sponge=None, heating=None):
    P = CompParams()
    P.nx, P.ny, P.ng = nx, ny, ng
    P.dx, P.dy, P.gamma = dx, dy, gamma
    P.limiter, P.use_flattening = limiter, use_flattening
    P.z0, P.z1, P.delta = z0, z1, delta
    P.cvisc, P.grav, P.small_dens = cvisc, grav, small_dens
    vb = comp_var_bcs(bcs)
    for n in range(4):
        for s in range(4):
            P.bc[n][s] = int(vb[n][s])
    P.avisc_xhi_interior = avisc_xhi_interior
    P.avisc_yhi_interior = avisc_yhi_interior
    P.riemann = {"HLLC": 0, "CGF": 1, "HLLC_lm": 2}[riemann]
    solid = [int(b in ("reflect", "reflect-even", "reflect-odd", "dirichlet")) for b in bcs]
    P.solid_xl, P.solid_xr, P.solid_yl, P.solid_yr = solid
    if sponge is not None:
        P.do_sponge = 1
        P.sponge_rho_begin, P.sponge_rho_full, P.sponge_timescale = sponge
    if heating is not None:      # (rate, profile (qx,qy) incl. ghost coordinates)
        rate, prof = heating
        P._heat_keep = np.ascontiguousarray(prof, dtype=np.float64)   # keep alive
        P.heat_rate = rate
        P.heat_prof = _p(P._heat_keep)
    return P


def comp_var_bcs(bcs):
    """per-variable BC codes for (dens, ener, xmom, ymom) given the four mesh
    boundary types; 'reflect' is even except for the normal momentum
    (simulation_null.py:99-112, compressible/simulation.py:216-226)."""
    out = np.zeros((4, 4), dtype=np.int32)
    for n in range(4):
        for s, b in enumerate(bcs):
            if b == "reflect":
                odd = (n == 2 and s < 2) or (n == 3 and s >= 2)
                out[n, s] = 2 if odd else 1
            else:
                out[n, s] = BC_CODES[b]
    return out


def set_scalar_pow(on):
    """evaluate the reference's scalar `x**2` with libm pow (what the
    interpreted shim run does) instead of x*x (what numba compiles)."""
    lib().orc_set_scalar_pow(int(on))


def comp_fill_bc(U, nx, ny, ng, bcs, gamma=1.4, grav=0.0, dy=0.0,
                 ambient=(0.0, 0.0, 0.0, 0.0)):
    """CellCenterData2d.fill_BC_all on the conserved state, including the
    hse / ambient user boundaries (compressible/BC.py)."""
    vb = np.ascontiguousarray(comp_var_bcs(bcs))
    amb = np.asarray(ambient, dtype=np.float64)
    f = lib().orc_comp_fill_bc
    f.restype = None
    f(_p(U), nx, ny, ng, vb.ctypes.data_as(C.POINTER(C.c_int)),
      C.c_double(gamma), C.c_double(grav), C.c_double(dy), _p(amb))
    return U


def comp_dt(U, nx, ny, ng, dx, dy, gamma, cfl):
    _ck(U)
    return lib().orc_comp_dt(_p(U), nx, ny, ng, C.c_double(dx),
                             C.c_double(dy), C.c_double(gamma),
                             C.c_double(cfl))


def comp_dt_geom(U, nx, ny, ng, geom, gamma, cfl):
    """method_compute_timestep with the Lx, Ly arrays of a curvilinear grid"""
    _ck(U)
    f = lib().orc_comp_dt_geom
    f.restype = C.c_double
    return f(_p(U), nx, ny, ng, geom.Lx, geom.Ly, C.c_double(gamma), C.c_double(cfl))


def comp_step(U, P, dt, stages=False, geom=None):
    """one compressible step in place on U (qx,qy,4) (ghosts filled).
    geom: a Geom (SphericalPolar grid) or None.  returns (rc, stages dict)"""
    _ck(U)
    P.geom = C.cast(C.pointer(geom), C.c_void_p) if geom is not None else None
    outs = {}
    st = CompStages()
    if stages:
        qx, qy = U.shape[:2]
        for name in _STAGE_NAMES:
            shp = (qx, qy) if name in _SCALAR_STAGES else (qx, qy, 4)
            outs[name] = np.zeros(shp)
            setattr(st, name, _p(outs[name]))
    rc = lib().orc_comp_step(_p(U), C.byref(P), C.c_double(dt),
                             C.byref(st) if stages else None)
    return rc, outs


class MG:
    """mirror of MG.CellCenterMG2d's numerical core (constant coefficients)"""

    def __init__(self, nx, xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0,
                 bcs=("dirichlet",) * 4, alpha=0.0, beta=-1.0, nsmooth=10,
                 nsmooth_bottom=50):
        bc = bc_codes(bcs)
        self._l = lib()
        self.h = C.c_void_p(self._l.orc_mg_create(
            nx, C.c_double(xmin), C.c_double(xmax), C.c_double(ymin),
            C.c_double(ymax), bc.ctypes.data_as(C.POINTER(C.c_int)),
            C.c_double(alpha), C.c_double(beta), nsmooth, nsmooth_bottom))
        self.nx = nx
        self.nlevels = self._l.orc_mg_nlevels(self.h)

    def __del__(self):
        try:
            self._l.orc_mg_free(self.h)
        except Exception:
            pass

    def arr(self, level, var):
        """numpy view of a level array; var: 0=v 1=f 2=r"""
        n = 2 ** (level + 1) + 2
        p = self._l.orc_mg_ptr(self.h, level, var)
        return np.ctypeslib.as_array(p, shape=(n, n))

    def set_bcval(self, side, vals):
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        self._l.orc_mg_set_bcval(self.h, side, _p(vals))

    def init_rhs(self, f):
        self.arr(self.nlevels - 1, 1)[:, :] = f
        self._l.orc_mg_init_rhs_norm(self.h)

    def smooth(self, level, n):
        self._l.orc_mg_smooth(self.h, level, n)

    def residual(self, level):
        self._l.orc_mg_residual(self.h, level)

    def restrict(self, level):
        self._l.orc_mg_restrict(self.h, level)

    def prolong_add(self, level):
        self._l.orc_mg_prolong_add(self.h, level)

    def fill_bc_v(self, level):
        self._l.orc_mg_fill_bc_v(self.h, level)

    def vcycle(self, level=None):
        self._l.orc_mg_vcycle(self.h, self.nlevels - 1 if level is None else level)

    def norm(self, level, var):
        return self._l.orc_mg_norm(self.h, level, var)

    def solve(self, rtol=1.e-11, max_cycles=100):
        self._l.orc_mg_set_max_cycles(self.h, max_cycles)
        self._l.orc_mg_solve(self.h, C.c_double(rtol))
        g = self._l.orc_mg_get_scalar
        self.source_norm = g(self.h, 0)
        self.num_cycles = int(g(self.h, 1))
        self.relative_error = g(self.h, 2)
        self.residual_error = g(self.h, 3)


class VCMG(MG):
    """variable-coefficient subclass (variable_coeff_MG.VarCoeffCCMG2d core)"""

    def __init__(self, nx, coeffs, xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0,
                 bcs=("dirichlet",) * 4, coeffs_bcs=("neumann",) * 4, nsmooth=10,
                 nsmooth_bottom=50):
        bc, cbc = bc_codes(bcs), bc_codes(coeffs_bcs)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
        self._l = lib()
        self.vh = C.c_void_p(self._l.orc_vcmg_create(
            nx, C.c_double(xmin), C.c_double(xmax), C.c_double(ymin), C.c_double(ymax),
            bc.ctypes.data_as(C.POINTER(C.c_int)), cbc.ctypes.data_as(C.POINTER(C.c_int)),
            _p(coeffs), nsmooth, nsmooth_bottom))
        self.h = C.c_void_p(self._l.orc_vcmg_base(self.vh))
        self.nx = nx
        self.nlevels = self._l.orc_mg_nlevels(self.h)

    def __del__(self):
        try:
            self._l.orc_vcmg_free(self.vh)
        except Exception:
            pass

    def coef(self, level, which):
        """0 = cell coefficient, 1 = eta_x, 2 = eta_y"""
        n = 2 ** (level + 1) + 2
        return np.ctypeslib.as_array(self._l.orc_vcmg_ptr(self.vh, level, which), shape=(n, n))

    def smooth(self, level, n):
        self._l.orc_vcmg_smooth(self.vh, level, n)

    def residual(self, level):
        self._l.orc_vcmg_residual(self.vh, level)

    def vcycle(self, level=None):
        self._l.orc_vcmg_vcycle(self.vh, self.nlevels - 1 if level is None else level)

    def solve(self, rtol=1.e-11, max_cycles=100):
        self._l.orc_mg_set_max_cycles(self.h, max_cycles)
        self._l.orc_vcmg_solve(self.vh, C.c_double(rtol))
        g = self._l.orc_mg_get_scalar
        self.source_norm = g(self.h, 0)
        self.num_cycles = int(g(self.h, 1))
        self.relative_error = g(self.h, 2)
        self.residual_error = g(self.h, 3)


# ---- burgers / incompressible (rows f1, f4) ------------------------------
def bg_edge_states(u, v, gpx, gpy, nx, ny, ng, dx, dy, dt, limiter):
    """8 planes u_xl,u_xr,u_yl,u_yr,v_xl,v_xr,v_yl,v_yr"""
    E = np.zeros((8,) + u.shape)
    lib().orc_bg_edge_states(_p(u), _p(v), None if gpx is None else _p(gpx),
                             None if gpy is None else _p(gpy), nx, ny, ng,
                             C.c_double(dx), C.c_double(dy), C.c_double(dt), limiter, _p(E))
    return E


def bg_dt(u, v, nx, ny, ng, dx, dy, cfl):
    f = lib().orc_bg_dt
    f.restype = C.c_double
    return f(_p(u), _p(v), nx, ny, ng, C.c_double(dx), C.c_double(dy), C.c_double(cfl))


def bg_step(u, v, nx, ny, ng, dx, dy, dt, limiter):
    lib().orc_bg_step(_p(u), _p(v), nx, ny, ng, C.c_double(dx), C.c_double(dy),
                      C.c_double(dt), limiter)


def _bc4(bcs):
    return np.ascontiguousarray(bc_codes(bcs))


def bgv_step(u, v, nx, ng, dt, limiter, eps, bc_u=("periodic",) * 4, bc_v=("periodic",) * 4,
             xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0):
    """one burgers_viscous evolve() in place on u, v (qx, qy), ghost cells filled;
    returns the V-cycle counts of the two Helmholtz solves"""
    bu, bv = _bc4(bc_u), _bc4(bc_v)
    ncyc = np.zeros(2, dtype=np.int32)
    lib().orc_bgv_step(_p(u), _p(v), nx, ng, C.c_double(xmin), C.c_double(xmax),
                       C.c_double(ymin), C.c_double(ymax), C.c_double(dt), limiter,
                       C.c_double(eps), bu.ctypes.data_as(C.POINTER(C.c_int)),
                       bv.ctypes.data_as(C.POINTER(C.c_int)),
                       ncyc.ctypes.data_as(C.POINTER(C.c_int)))
    return tuple(int(x) for x in ncyc)


def incomp_step(D, nx, ng, dt, limiter=2, proj_type=2, bc_u=("periodic",) * 4,
                bc_v=("periodic",) * 4, bc_phi=("periodic",) * 4,
                xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0, stages=False):
    """one incompressible evolve() on D = (6, qx, qy) planes u, v, phi-MAC,
    phi, gradp_x, gradp_y.  Returns dict(ncyc=..., [umac, vmac, adv])"""
    _ck(D)
    N = D.shape[1:]
    um = np.zeros(N) if stages else None
    vm = np.zeros(N) if stages else None
    adv = np.zeros((2,) + N) if stages else None
    ncyc = (C.c_int * 4)()
    bu, bv, bp = _bc4(bc_u), _bc4(bc_v), _bc4(bc_phi)
    ip = C.POINTER(C.c_int)
    lib().orc_incomp_step(_p(D), nx, ng, C.c_double(xmin), C.c_double(xmax), C.c_double(ymin),
                          C.c_double(ymax), C.c_double(dt), limiter, proj_type,
                          bu.ctypes.data_as(ip), bv.ctypes.data_as(ip), bp.ctypes.data_as(ip), 0,
                          None if um is None else _p(um), None if vm is None else _p(vm),
                          None if adv is None else _p(adv), ncyc)
    out = {"ncyc": (ncyc[0], ncyc[1]), "ncyc_visc": (ncyc[2], ncyc[3])}
    if stages:
        out.update(umac=um, vmac=vm, adv=adv)
    return out


def incomp_set_viscous(nu=None, lid_u=1.0, lid_v=0.0):
    """nu = None: inviscid incompressible solver; else incompressible_viscous"""
    lib().orc_incomp_set_viscous(int(nu is not None), C.c_double(0.0 if nu is None else nu),
                                 C.c_double(lid_u), C.c_double(lid_v))


def incomp_preevolve(D, nx, ng, cfl, limiter=2, proj_type=2, bc_u=("periodic",) * 4,
                     bc_v=("periodic",) * 4, bc_phi=("periodic",) * 4,
                     xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0):
    _ck(D)
    bu, bv, bp = _bc4(bc_u), _bc4(bc_v), _bc4(bc_phi)
    ip = C.POINTER(C.c_int)
    f = lib().orc_incomp_preevolve
    f.restype = C.c_double
    return f(_p(D), nx, ng, C.c_double(xmin), C.c_double(xmax), C.c_double(ymin),
             C.c_double(ymax), C.c_double(cfl), limiter, proj_type,
             bu.ctypes.data_as(ip), bv.ctypes.data_as(ip), bp.ctypes.data_as(ip))


# ---- compressible_rk (row f4) ---------------------------------------------
def comp_rk_rhs(U, P, fluxes=False):
    """k = -div F + S of the stage state U (qx,qy,4), ghost cells filled; U's
    density is floored in place.  Returns (rc, k[, Fx, Fy])"""
    _ck(U)
    k = np.zeros_like(U)
    Fx = np.zeros_like(U) if fluxes else None
    Fy = np.zeros_like(U) if fluxes else None
    rc = lib().orc_comp_rk_rhs(_p(U), C.byref(P), _p(k), None if Fx is None else _p(Fx),
                               None if Fy is None else _p(Fy))
    return (rc, k, Fx, Fy) if fluxes else (rc, k)


def comp_rk_dt(U, nx, ny, ng, dx, dy, gamma, cfl):
    f = lib().orc_comp_rk_dt
    f.restype = C.c_double
    return f(_p(U), nx, ny, ng, C.c_double(dx), C.c_double(dy), C.c_double(gamma),
             C.c_double(cfl))


# ---- shallow water (row f4) -------------------------------------------------
class SweParams(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("ng", C.c_int), ("dx", C.c_double),
                ("dy", C.c_double), ("g", C.c_double), ("limiter", C.c_int),
                ("riemann", C.c_int)]


class SweStages(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_double)) for n in
                ("Uxl0", "Uxr0", "Uyl0", "Uyr0", "FxT", "FyT", "Fx", "Fy")]


def swe_params(nx, ny, ng, dx, dy, g, limiter, riemann):
    return SweParams(int(nx), int(ny), int(ng), dx, dy, g, int(limiter),
                     {"Roe": 0, "HLLC": 1}[riemann])


def swe_dt(U, P, cfl):
    f = lib().orc_swe_dt
    f.restype = C.c_double
    return f(_p(U), P.nx, P.ny, P.ng, C.c_double(P.dx), C.c_double(P.dy), C.c_double(P.g),
             C.c_double(cfl))


def swe_step(U, P, dt, stages=False):
    _ck(U)
    st = SweStages()
    outs = {}
    if stages:
        for n, _t in SweStages._fields_:
            outs[n] = np.zeros_like(U)
            setattr(st, n, _p(outs[n]))
    lib().orc_swe_step(_p(U), C.byref(P), C.c_double(dt), C.byref(st) if stages else None)
    return outs


class GenMG(VCMG):
    """general_MG.GeneralMG2d core: alpha phi + div(beta grad phi) + gamma . grad phi = f.
    coef(level, which): 0 beta, 1 beta_x, 2 beta_y, 3 alpha, 4 gamma_x, 5 gamma_y"""

    def __init__(self, nx, alpha, beta, gamma_x, gamma_y, xmin=0.0, xmax=1.0, ymin=0.0, ymax=1.0,
                 bcs=("dirichlet",) * 4, coeffs_bcs=("neumann",) * 4, nsmooth=10,
                 nsmooth_bottom=50):
        bc = bc_codes(bcs)
        cb = np.ascontiguousarray(np.tile(bc_codes(coeffs_bcs), 4))
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (alpha, beta, gamma_x, gamma_y)]
        self._l = lib()
        self._l.orc_genmg_create.restype = C.c_void_p
        ip = C.POINTER(C.c_int)
        self.vh = C.c_void_p(self._l.orc_genmg_create(
            nx, C.c_double(xmin), C.c_double(xmax), C.c_double(ymin), C.c_double(ymax),
            bc.ctypes.data_as(ip), cb.ctypes.data_as(ip), *[_p(a) for a in arrs],
            nsmooth, nsmooth_bottom))
        self.h = C.c_void_p(self._l.orc_vcmg_base(self.vh))
        self.nx = nx
        self.nlevels = self._l.orc_mg_nlevels(self.h)


# ---- "ramp" boundary of the double Mach reflection problem ------------------
def ramp_params(nx, ny, ng, xmin, xmax, ymin, ymax, gamma, t):
    """everything compressible/BC.py:178-296 evaluates with math.*: cell centres,
    sub-sampling offset, inflow states (inflow_post_bc / inflow_pre_bc, :254-296,
    in the state's order density, energy, x-momentum, y-momentum) and the shock
    front of the ghost rows above the upper y boundary (:240-243) at time t"""
    import math
    dx, dy = (xmax - xmin) / nx, (ymax - ymin) / ny
    qx, qy = nx + 2 * ng, ny + 2 * ng
    xl = (np.arange(qx) - ng) * dx + xmin
    xr = (np.arange(qx) + 1.0 - ng) * dx + xmin
    x = 0.5 * (xl + xr)
    yl = (np.arange(qy) - ng) * dy + ymin
    yr = (np.arange(qy) + 1.0 - ng) * dy + ymin
    y = 0.5 * (yl + yr)
    r_l, u_l, v_l, p_l = 8.0, 7.1447096, -4.125, 116.5
    r_r, u_r, v_r, p_r = 1.4, 0.0, 0.0, 1.0
    post = np.array([r_l, p_l / (gamma - 1.0) + 0.5 * r_l * (u_l * u_l + v_l * v_l), r_l * u_l,
                     r_l * v_l])
    pre = np.array([r_r, p_r / (gamma - 1.0) + 0.5 * r_r * (u_r * u_r + v_r * v_r), r_r * u_r,
                    r_r * v_r])
    jhi = ng + ny - 1
    sfd, sfu = np.zeros(8), np.zeros(8)
    for k in range(ng):
        yj = float(y[jhi + 1 + k])
        sfu[k] = 1.0 / 6.0 + (yj + 0.5 * dy * math.sqrt(3)) / math.tan(math.pi / 3.0) + \
            (10.0 / math.sin(math.pi / 3.0)) * t
        sfd[k] = 1.0 / 6.0 + (yj - 0.5 * dy * math.sqrt(3)) / math.tan(math.pi / 3.0) + \
            (10.0 / math.sin(math.pi / 3.0)) * t
    return dict(x=np.ascontiguousarray(x), cxoff=0.5 * dx * math.sqrt(3), post=post, pre=pre,
                sf_down=sfd, sf_up=sfu)


def comp_fill_bc_ramp(U, nx, ny, ng, bcs, rp):
    vb = np.ascontiguousarray(comp_var_bcs(bcs))
    f = lib().orc_comp_fill_bc_ramp
    f.restype = None
    f(_p(U), nx, ny, ng, vb.ctypes.data_as(C.POINTER(C.c_int)), _p(rp["x"]),
      C.c_double(rp["cxoff"]), _p(rp["post"]), _p(rp["pre"]), _p(rp["sf_down"]), _p(rp["sf_up"]))
