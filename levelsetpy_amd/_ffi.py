"""ctypes binding of libhj_mi355x.so (include/hj_mi355x.h).

The product path is HIP only: if the shared library is missing or cannot be
loaded, every compute entry point raises -- there is no CPU fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HJ_LIB") or os.path.join(HERE, "csrc", "libhj_mi355x.so")   # HJ_LIB: tuning builds

# enums of include/hj_mi355x.h
BC_EXTRAPOLATE, BC_PERIODIC = 0, 1
DISS_GLF, DISS_LOCAL = 0, 1
DISS_LLF, DISS_LLLF = 1, 2          # distinguished by run-time Hamiltonians that read the costate range (hj_mi355x.h)
POST_NONE, POST_MIN_PREV, POST_MAX_PREV = 0, 1, 2
ENO2, ENO3, WENO5, WENO5_ASSHIPPED = 0, 1, 2, 3
HAM_DUBINS_REL, HAM_DOUBLE_INTEGRATOR, HAM_DOUBLE_PENDULUM = 0, 1, 2
HAM_USER_BASE = 100
F64, F32 = 0, 1
STAGE_YDOT, STAGE_EULER, STAGE_RK3_HALF, STAGE_RK3_FULL, STAGE_RK2_FULL = 0, 1, 2, 3, 4
OP_MIN, OP_MAX, OP_MAX_NEG = 0, 1, 2
STENCIL = 3
BOUND_SLOTS = 64

SCHEME_IDS = {"ENO2": ENO2, "ENO3": ENO3, "WENO5": WENO5, "WENO5_ASSHIPPED": WENO5_ASSHIPPED, "ENO2_FAST": 4, "ENO3_FAST": 5}

_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_pd, _pi, _pi64 = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int64)

# name -> (restype, argtypes): every symbol the header declares
SIGNATURES = {
    "hj_ctx_create": (_i, [C.POINTER(_vp), _i, _pi64, _pd, _pd, _pi, _pi, _i, _i]),
    "hj_ctx_destroy": (None, [_vp]),
    "hj_ctx_set_stream": (_i, [_vp, _vp]),
    "hj_ctx_set_coords": (_i, [_vp, _i, _pd]),
    "hj_ctx_set_aux": (_i, [_vp, _i, _pd, _i64]),
    "hj_ctx_set_slab": (_i, [_vp, _i, _i]),
    "hj_ghost": (_i, [_vp, _i, _i, _vp, _vp]),
    "hj_upwind": (_i, [_vp, _i, _i, _vp, _vp, _vp, _pd]),
    "hj_lf_split_begin": (_i, [_vp, _i, _vp, C.POINTER(_vp), C.POINTER(_vp), _pd]),
    "hj_lf_split_end": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _pd, _vp, _vp, _pd, _pd]),
    "hj_rk_combine": (_i, [_vp, _i, _d, _vp, _vp, _vp, _vp, _i64]),
    "hj_lf_term": (_i, [_vp, _i, _i, _pd, _d, _i, _vp, _vp, _pd]),
    "hj_rk_substep": (_i, [_vp, _i, _i, _pd, _d, _i, _d, _i, _vp, _vp, _vp, _i, _i64, _i64]),
    "hj_read_step_bound": (_i, [_vp, _i, _pd, _pd]),
    "hj_rk_step": (_i, [_vp, _i, _i, _i, _pd, _d, _d, _d, _d, _i, _vp, _vp, _vp, _vp, _pd, _pd]),
    "hj_rk_stage12": (_i, [_vp, _i, _i, _pd, _d, _d, _d, _vp, _vp, _i]),
    "hj_rk_plan": (_i, [_vp, _i, _i, _i, _pd, _i, _pi, _pi]),
    "hj_rk_integrate": (_i, [_vp, _i, _i, _i, _pd, _d, _d, _d, _d, _i, _vp, _vp, _vp, _vp, _i64, _d, _pd, _pi64, _pi]),
    "hj_ctx_set_post_step": (_i, [_vp, _i]),
    "hj_ctx_set_post_arrays": (_i, [_vp, _i, _vp, _i, _vp]),
    "hj_static_step_bound": (_i, [_vp, _i, _pd, _pd, _pd]),
    "hj_ctx_set_dissipation": (_i, [_vp, _i]),
    "hj_max_d1sq": (_i, [_vp, _vp, _vp]),
    "hj_ctx_set_weno_eps_source": (_i, [_vp, _vp]),
    "hj_minmax_with": (_i, [_vp, _i, _vp, _vp, _i64]),
    "hj_any_nan": (_i, [_vp, _vp, _i64, _pi]),
    "hj_comm_unique_id": (_i, [C.c_char_p, _vp]),
    "hj_comm_init": (_i, [_vp, C.c_char_p, _i, _i, _vp, _i, _i]),
    "hj_comm_destroy": (_i, [_vp]),
    "hj_comm_info": (_i, [_vp, _pi, _pi, _pi, _pi]),
    "hj_halo_exchange": (_i, [_vp, _vp]),
    "hj_slab_join": (_i, [_vp]),
    "hj_slab_rk_step": (_i, [_vp, _i, _i, _i, _pd, _d, _i, _vp, _vp, _vp, _vp]),
    "hj_ctx_set_axis0_pad": (_i, [_vp, _i, _pd, _pd, _pd]),
    "hj_comm_init_external": (_i, [_vp, _i, _i, _i, _i]),
    "hj_halo_exchange_depth": (_i, [_vp, _vp, _i]),
    "hj_slab_rk_step_deep": (_i, [_vp, _i, _i, _i, _pd, _d, _i, _vp, _vp, _vp, _vp]),
    "hj_term_normal": (_i, [_vp, _i, _vp, _vp, _d, _vp, _pd]),
    "hj_term_reinit": (_i, [_vp, _i, _vp, _vp, _i, _vp, _pd]),
    "hj_term_convection": (_i, [_vp, _i, _vp, _vp, _pd, _vp, _pd]),
    "hj_ham_register": (_i, [C.c_char_p, _i, _i, C.c_char_p, C.c_char_p, _i, C.c_char_p, C.c_char_p, _pi]),
    "hj_ham_register2": (_i, [C.c_char_p, _i, _i, C.c_char_p, C.c_char_p, _i, _i, C.c_char_p, C.c_char_p, _pi]),
    "hj_ham_flags": (_i, [_i, _pi]),
    "hj_rk_last_bounds": (_i, [_vp, _pd, _pi]),
    "hj_rk_prev_bounds": (_i, [_vp, _pd, _pi, _pd]),
    "hj_range_pass": (_i, [_vp, _i, _i, _pd, _vp, _vp]),
    "hj_bound_pass": (_i, [_vp, _i, _i, _pd, _vp, _pd]),
    "hj_ctx_set_range_source": (_i, [_vp, _vp]),
    "hj_range_alpha_max": (_i, [_vp, _i, _pd, _pd]),
    "hj_ham_info": (_i, [_i, _pi, _pi, _pi]),
    "hj_ham_compile_check": (_i, [_i, _i]),
    "hj_ham_cache_stats": (_i, [_pi, _pi]),
    "hj_sync": (_i, [_vp]),
    "hj_last_error": (C.c_char_p, []),
    "hj_last_kernel": (C.c_char_p, [C.c_void_p]),
    "hj_last_launch": (_i, [_vp, _pi, _pi]),
    "hj_last_tile": (_i, [_vp, _pi]),
    "hj_ctx_state_generation": (C.c_uint64, [_vp]),
    "hj_plan_substep": (_i, [_i, _pi64, _pi, _i, _i, _i, _i, _i64, _i64, _i, _i, _i, _pi64, C.c_char_p, _i]),
    "hj_version": (C.c_char_p, []),
}

HAM_RANGE = 1          # hj_ham_register2 flag: alpha reads the costate range (dmin / dmax)

_lib = None


def lib():
    """The loaded library; raises RuntimeError (loudly) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "levelsetpy_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C levelsetpy_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        # torch first: its wheel bundles the HIP/HSA runtime the process must share.  Loaded the other way
        # round, the library would bring in /opt/rocm's copy and torch a second one, and the later of the
        # two runtimes to initialise finds "no ROCm-capable device" (seen with build() followed by smoke()).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class Unsupported(ValueError):
    """HJ_EUNSUPPORTED: the library has no kernel for this combination (callers with another path may take it)."""


def check(rc):
    """Non-zero return code -> ValueError, as the reference's error() (matlab_utils.py:134-137)."""
    if rc != 0:
        msg = lib().hj_last_error()
        text = (msg or b"hj_mi355x error").decode("utf-8", "replace") + " (code %d)" % rc
        raise (Unsupported if rc == -3 else ValueError)(text)


def darr(values):
    values = [float(v) for v in values]
    return (C.c_double * max(1, len(values)))(*values)
