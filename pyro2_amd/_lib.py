"""ctypes binding of libpyrohip.so (include/pyrohip.h).

The product path has NO CPU fallback: if the HIP library cannot be loaded, or
it is not the gfx950 build, importing a device object fails loudly.  (Tests
that run on the GPU-less build container inject the host-emulated build of the
same kernel sources explicitly through `use_library`.)
"""
import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "lib", "libpyrohip.so")

BC_OUTFLOW, BC_REFLECT_EVEN, BC_REFLECT_ODD, BC_PERIODIC, BC_HALO, BC_HSE, BC_AMBIENT, BC_RAMP, \
    BC_CONST = range(9)
BC_CODE = {"outflow": BC_OUTFLOW, "neumann": BC_OUTFLOW,
           "reflect-even": BC_REFLECT_EVEN, "reflect-odd": BC_REFLECT_ODD,
           "dirichlet": BC_REFLECT_ODD, "periodic": BC_PERIODIC,
           "halo": BC_HALO, "hse": BC_HSE, "ambient": BC_AMBIENT, "ramp": BC_RAMP,
           "moving_lid": BC_CONST}

class GeomArrays(C.Structure):
    """pyrohip_geom: the arrays of a SphericalPolar grid (include/pyrohip.h)"""
    NAMES = ("Lx", "Ly", "Ax", "Ay", "V", "dlogAx", "dlogAy", "x2d", "sint", "sinb", "sinc")
    _fields_ = [(n, C.POINTER(C.c_double)) for n in NAMES] + \
               [("xmin", C.c_double), ("ymin", C.c_double),
                ("rowf", C.POINTER(C.c_double)), ("colf", C.POINTER(C.c_double))]


ERR_STATE = 10002
UNIQUE_ID_BYTES = 128


class PyroHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libpyrohip error {code}: {msg}")
        self.code = code


class AdvParams(C.Structure):
    _fields_ = [("dx", C.c_double), ("dy", C.c_double), ("u", C.c_double), ("v", C.c_double),
                ("limiter", C.c_int), ("fill", C.c_int), ("fast_math", C.c_int),
                ("march_rows", C.c_int), ("multi_k", C.c_int), ("multi_prio", C.c_int)]


class MGTuning(C.Structure):
    _fields_ = [("kmax", C.c_int), ("kmax_small", C.c_int), ("nsmall", C.c_int),
                ("march_min", C.c_int), ("march_waves", C.c_int), ("march_side", C.c_double),
                ("march_minrows", C.c_int), ("fuse_res_restrict", C.c_int),
                ("lazy_residual", C.c_int), ("allow_pow2", C.c_int), ("small_tiles", C.c_int),
                ("band_maxn", C.c_int), ("band_genedge", C.c_int), ("coarse_band64", C.c_int),
                ("speculate", C.c_int), ("trace", C.c_int), ("spec_debug", C.c_int),
                ("march_tail", C.c_int), ("coarse_wave", C.c_int)]


class CompParams(C.Structure):
    _fields_ = [("dx", C.c_double), ("dy", C.c_double), ("gamma", C.c_double),
                ("limiter", C.c_int), ("use_flattening", C.c_int),
                ("z0", C.c_double), ("z1", C.c_double), ("delta", C.c_double),
                ("cvisc", C.c_double), ("grav", C.c_double),
                ("small_dens", C.c_double),
                ("avisc_xhi_interior", C.c_int),
                ("avisc_yhi_interior", C.c_int),
                ("fast_math", C.c_int), ("kernel_set", C.c_int),
                ("riemann", C.c_int), ("solid_xl", C.c_int), ("solid_yl", C.c_int),
                ("do_sponge", C.c_int), ("sponge_rho_begin", C.c_double),
                ("sponge_rho_full", C.c_double), ("sponge_timescale", C.c_double),
                ("heat_rate", C.c_double), ("march_rows", C.c_int), ("fuse_fill", C.c_int),
                ("step_launches", C.c_int)]


class DtPolicyC(C.Structure):
    """pyrohip_dt_policy (include/pyrohip.h)"""
    _fields_ = [("tmax", C.c_double), ("init_tstep_factor", C.c_double),
                ("max_dt_change", C.c_double), ("fix_dt", C.c_double),
                ("t", C.c_double), ("dt_old", C.c_double), ("n", C.c_longlong)]


_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int)
_VP = C.c_void_p

_PROTOS = {
    "pyrohip_init": [C.c_int, C.POINTER(_VP)],
    "pyrohip_shutdown": [_VP],
    "pyrohip_sync": [_VP],
    "pyrohip_device_info": [_VP, C.c_char_p, C.c_int, C.POINTER(C.c_size_t),
                            C.POINTER(C.c_size_t), _IP],
    "pyrohip_prof_enable": [_VP, C.c_int],
    "pyrohip_prof_report": [_VP, C.c_char_p, C.c_int],
    "pyrohip_timer_start": [_VP],
    "pyrohip_timer_stop": [_VP, _DP],
    "pyrohip_state_create": [_VP, C.c_int, C.c_int, C.c_int, C.c_int, _IP,
                             C.POINTER(_VP)],
    "pyrohip_state_destroy": [_VP],
    "pyrohip_state_upload": [_VP, _DP],
    "pyrohip_state_download": [_VP, _DP],
    "pyrohip_state_upload_var": [_VP, C.c_int, _DP],
    "pyrohip_state_download_var": [_VP, C.c_int, _DP],
    "pyrohip_state_upload_rows": [_VP, C.c_int, C.c_int, _DP],
    "pyrohip_state_download_rows": [_VP, C.c_int, C.c_int, _DP],
    "pyrohip_device_count": [C.POINTER(C.c_int)],
    "pyrohip_comm_set_global_dt": [_VP, C.c_int],
    "pyrohip_comp_dt_is_global": [_VP, C.POINTER(C.c_int)],
    "pyrohip_comp_rk_dt_is_cached": [_VP, C.POINTER(C.c_int)],
    "pyrohip_comp_dt_is_cached": [_VP, C.POINTER(C.c_int)],
    "pyrohip_mg_set_general_coeffs": [_VP, _DP, _DP, _DP, _DP, C.POINTER(C.c_int)],
    "pyrohip_comp_rk_rhs": [_VP, C.POINTER(CompParams), _VP, C.c_int],
    "pyrohip_comp_rk_dt": [_VP, C.POINTER(CompParams), C.c_double, _DP],
    "pyrohip_comp_rk_can_fuse": [_VP, C.POINTER(CompParams), _VP, C.c_int, C.POINTER(C.c_int)],
    "pyrohip_comp_rk_step": [_VP, C.POINTER(CompParams), _VP, C.c_double, C.c_int, _DP, _DP],
    "pyrohip_comp_rk_evolve": [_VP, C.POINTER(CompParams), _VP, C.c_int, _DP, _DP, C.c_double,
                               C.POINTER(DtPolicyC), C.c_int, C.POINTER(C.c_int), _DP],
    "pyrohip_state_lincomb": [_VP, _VP, _VP, _DP, C.c_int],
    "pyrohip_swe_dt": [_VP, C.c_double, C.c_double, C.c_double, C.c_double, _DP],
    "pyrohip_swe_step": [_VP, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double],
    "pyrohip_swe_step_ks": [_VP, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int],
    "pyrohip_swe_step_ex": [_VP, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int,
                            C.c_int],
    "pyrohip_swe_evolve": [_VP, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double,
                           C.POINTER(DtPolicyC), C.c_int, C.POINTER(C.c_int), _DP],
    "pyrohip_swe_stage_dump": [_VP, C.c_int, _DP],
    "pyrohip_bg_step": [_VP, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int],
    "pyrohip_inc_mac_rhs": [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                            C.c_double, C.c_double, C.c_int, C.c_double, _DP],
    "pyrohip_inc_visc_rhs": [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                             C.c_double, C.c_double, C.c_int, _DP],
    "pyrohip_inc_visc_store": [_VP, _VP, C.c_int],
    "pyrohip_bgv_predict": [_VP, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                            C.c_double],
    "pyrohip_bgv_rhs": [_VP, _VP, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                        _DP],
    "pyrohip_state_set_const_bc": [_VP, C.c_int, C.c_double],
    "pyrohip_state_set_geometry": [_VP, C.c_void_p],
    "pyrohip_mg_set_helmholtz": [_VP, C.c_double, C.c_double],
    "pyrohip_inc_advect": [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                           C.c_double, C.c_double, C.c_int],
    "pyrohip_inc_proj_rhs": [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                             C.c_double, C.c_int, _DP],
    "pyrohip_inc_proj_update": [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_double, C.c_double, C.c_double, C.c_int],
    "pyrohip_inc_stage_dump": [_VP, C.c_int, _DP],
    "pyrohip_fill_bc": [_VP, C.c_int],
    "pyrohip_state_set_user_bc": [_VP, C.c_double, C.c_double, C.c_double, _DP],
    "pyrohip_state_set_heating": [_VP, _DP],
    "pyrohip_state_set_source": [_VP, C.c_int, _VP],
    "pyrohip_comp_source_correct": [_VP, C.POINTER(CompParams), C.c_double],
    "pyrohip_state_set_ramp_bc": [_VP, _DP, C.c_double, _DP, _DP, _DP, _DP],
    "pyrohip_state_minmax": [_VP, C.c_int, C.c_int, _DP, _DP],
    "pyrohip_adv_step": [_VP, C.c_int, C.c_double, C.c_double, C.c_double,
                         C.c_double, C.c_double, C.c_int],
    "pyrohip_adv_step_fill": [_VP, C.c_int, C.c_double, C.c_double, C.c_double,
                              C.c_double, C.c_double, C.c_int, C.c_int],
    "pyrohip_adv_step_p": [_VP, C.c_int, C.POINTER(AdvParams), C.c_double],
    "pyrohip_comp_wave_geometry": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)],
    "pyrohip_adv_evolve": [_VP, C.c_int, C.POINTER(AdvParams), C.POINTER(C.c_double), C.c_int],
    "pyrohip_comp_dt": [_VP, C.POINTER(CompParams), C.c_double, _DP],
    "pyrohip_comp_evolve": [_VP, C.POINTER(CompParams), C.c_double, C.POINTER(DtPolicyC), C.c_int,
                            _IP, _DP],
    "pyrohip_comp_step": [_VP, C.POINTER(CompParams), C.c_double],
    "pyrohip_comp_stage_dump": [_VP, C.c_int, _DP],
    "pyrohip_mg_create": [_VP, C.c_int, C.c_double, C.c_double, C.c_double,
                          C.c_double, _IP, C.c_double, C.c_double, C.c_int,
                          C.c_int, C.POINTER(_VP)],
    "pyrohip_mg_destroy": [_VP],
    "pyrohip_mg_nlevels": [_VP, _IP],
    "pyrohip_mg_set_smoother": [_VP, C.c_int],
    "pyrohip_mg_set": [_VP, C.c_int, C.c_int, _DP],
    "pyrohip_mg_get": [_VP, C.c_int, C.c_int, _DP],
    "pyrohip_mg_set_bcval": [_VP, C.c_int, _DP],
    "pyrohip_mg_zero": [_VP, C.c_int, C.c_int],
    "pyrohip_mg_fill_bc": [_VP, C.c_int, C.c_int],
    "pyrohip_mg_smooth": [_VP, C.c_int, C.c_int],
    "pyrohip_mg_smooth_rows": [_VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
    "pyrohip_mg_residual_restrict_rows": [_VP, C.c_int, C.c_int, C.c_int],
    "pyrohip_mg_get_rows": [_VP, C.c_int, C.c_int, C.c_int, C.c_int, _DP],
    "pyrohip_mg_set_rows": [_VP, C.c_int, C.c_int, C.c_int, C.c_int, _DP],
    "pyrohip_mg_mark_zero": [_VP, C.c_int],
    "pyrohip_mg_exchange_rows": [_VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
    "pyrohip_mg_send_rows": [_VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
    "pyrohip_mg_recv_rows": [_VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
    "pyrohip_comm_group": [C.c_int],
    "pyrohip_mg_residual": [_VP, C.c_int],
    "pyrohip_mg_restrict": [_VP, C.c_int],
    "pyrohip_mg_prolong_add": [_VP, C.c_int],
    "pyrohip_mg_norm": [_VP, C.c_int, C.c_int, _DP],
    "pyrohip_mg_vcycle": [_VP, C.c_int],
    "pyrohip_mg_init_rhs_norm": [_VP, _DP],
    "pyrohip_mg_solve": [_VP, C.c_double, C.c_int, _IP, _DP, _DP],
    "pyrohip_mg_set_coeffs": [_VP, _DP, _IP],
    "pyrohip_mg_set_rhs_cn": [_VP, _VP, C.c_int, C.c_double, _DP],
    "pyrohip_mg_copy_solution": [_VP, _VP, C.c_int],
    "pyrohip_mg_get_tuning": [_VP, C.POINTER(MGTuning)],
    "pyrohip_mg_set_tuning": [_VP, C.POINTER(MGTuning)],
    "pyrohip_mg_tail_counts": [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "pyrohip_comm_unique_id": [C.c_char_p],
    "pyrohip_comm_init": [_VP, C.c_int, C.c_int, C.c_char_p],
    "pyrohip_comm_destroy": [_VP],
    "pyrohip_comm_size": [_VP, C.POINTER(C.c_int)],
    "pyrohip_halo_exchange": [_VP, C.c_int, C.c_int],
    "pyrohip_state_set_neighbours": [_VP, C.c_int, C.c_int],
    "pyrohip_state_send_rows": [_VP, C.c_int, C.c_int, C.c_int],
    "pyrohip_state_recv_rows": [_VP, C.c_int, C.c_int, C.c_int],
    "pyrohip_state_halo_pending": [_VP, _IP],
    "pyrohip_allreduce_min": [_VP, _DP],
    "pyrohip_allreduce_max": [_VP, _DP],
    "pyrohip_allreduce_sum": [_VP, _DP, C.c_int],
    "pyrohip_mg_rows_kmax": [_VP, C.c_int, _IP],
    "pyrohip_mg_diag_rows": [_VP, C.c_int, C.c_int, _DP],
    "pyrohip_mg_save_old": [_VP],
}

EXPORTS = sorted(list(_PROTOS) + ["pyrohip_last_error", "pyrohip_backend"])

_lock = threading.Lock()
_lib = None
_lib_path = None
_allow_backend = ("hip-gfx950",)


def use_library(path, allow_backends=("hip-gfx950",)):
    """Select the shared library to bind (call before the first device op).
    Only tests pass anything but the defaults."""
    global _lib, _lib_path, _allow_backend
    with _lock:
        _lib = None
        _lib_path = path
        _allow_backend = tuple(allow_backends)


def _load():
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        # PYRO2_AMD_LIB: developer override to A/B differently built gfx950
        # libraries; the backend check below still applies
        path = _lib_path or os.environ.get("PYRO2_AMD_LIB") or DEFAULT_LIB
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: the HIP extension has not been built "
                "(run `python -m pyro2_amd.build` or __graft_entry__.build()). "
                "pyro2_amd has no CPU fallback.")
        # kernel arguments in device memory instead of host-coherent memory: the first
        # scalar load of every workgroup otherwise crosses to the host (~1 us per launch,
        # 15-30 us per multigrid V-cycle, tools/mg_ab.sh).  Read by the HIP runtime when it
        # initialises; a value set by the user wins.
        os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
        lib = C.CDLL(path)
        lib.pyrohip_last_error.restype = C.c_char_p
        lib.pyrohip_backend.restype = C.c_char_p
        backend = lib.pyrohip_backend().decode()
        if backend not in _allow_backend:
            raise ImportError(f"{path} reports backend '{backend}', expected "
                              f"one of {_allow_backend}")
        for name, args in _PROTOS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        _lib = lib
        return lib


def lib():
    return _lib if _lib is not None else _load()


def backend_name():
    return lib().pyrohip_backend().decode()


def check(rc):
    if rc != 0:
        raise PyroHipError(rc, lib().pyrohip_last_error().decode())


def dptr(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], \
        "need a C-contiguous float64 array"
    return a.ctypes.data_as(_DP)


def iptr(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_IP)
