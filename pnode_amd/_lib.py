"""ctypes binding of the C ABI in ``include/pnode_amd.h`` (``pnode_amd/lib/libpnode_amd.so``).

The library is the product: if it is missing or a symbol is absent the import fails loudly --
there is no fallback path.  ``import torch`` happens first on purpose: the wheel bundles its
own HIP runtime (``torch/lib/libamdhip64.so``, soname ``libamdhip64.so.7``) and the loader
must resolve this library's dependency to that already-loaded runtime, otherwise the stream
handles torch hands us would belong to a different runtime instance.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpnode_amd.so")

PN_MAX_STAGES = 7
PN_MAX_TERMS = 8
PN_F32, PN_F64 = 0, 1
PN_TRAJ_ALL, PN_TRAJ_SOLUTION, PN_TRAJ_BUDGET = 0, 1, 2
KERNEL_IDS = ("pn_rk_stage", "pn_rk_combine_wrms", "pn_adj_theta", "pn_adj_accum", "pn_param_accum", "pn_copy", "pn_dots", "pn_lincomb",
              "pn_linear_wgrad")


class PnError(RuntimeError):
    """A C-ABI entry point returned non-zero (stands where petsc4py.PETSc.Error did)."""


class Tableau(ctypes.Structure):
    _fields_ = [
        ("s", ctypes.c_int), ("order", ctypes.c_int), ("fsal", ctypes.c_int), ("has_embed", ctypes.c_int),
        ("A", (ctypes.c_double * PN_MAX_STAGES) * PN_MAX_STAGES),
        ("b", ctypes.c_double * PN_MAX_STAGES),
        ("bembed", ctypes.c_double * PN_MAX_STAGES),
        ("c", ctypes.c_double * PN_MAX_STAGES),
    ]


class WgradPair(ctypes.Structure):
    """pn_wgrad_pair: one (cotangent, input) pair of pn_linear_wgrad_group."""
    _fields_ = [("g", ctypes.c_void_p), ("x", ctypes.c_void_p), ("pw", ctypes.c_void_p), ("pb", ctypes.c_void_p),
                ("alpha", ctypes.c_double), ("out_f", ctypes.c_int64), ("in_f", ctypes.c_int64)]


PN_WGRAD_MAX_PAIRS = 8
PN_WGRAD_EXACT_FP32 = 1
PN_WGRAD_TILE_64 = 2
PN_ABI_VERSION = 4
_vp, _i, _i64, _d, _cp = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_char_p
_pd, _pi, _pi64 = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64)
_pvp = ctypes.POINTER(ctypes.c_void_p)

# section 3a of the header: the callbacks of the step loops and the table of vector operations they launch
STAGE_CB = ctypes.CFUNCTYPE(ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_double)
VJP_CB = ctypes.CFUNCTYPE(ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double)
RK_STAGE_FN = ctypes.CFUNCTYPE(_i, _vp, _i, _i64, _vp, _vp, _i, _pvp, _pd)
RK_COMBINE_WRMS_FN = ctypes.CFUNCTYPE(_i, _vp, _i, _i64, _vp, _vp, _i, _pvp, _pd, _pd, _d, _d, _vp, _vp)
ADJ_THETA_FN = ctypes.CFUNCTYPE(_i, _vp, _i, _i64, _vp, _vp, _d, _i, _pvp, _pd)
ADJ_ACCUM_FN = ctypes.CFUNCTYPE(_i, _vp, _i, _i64, _vp, _vp, _i, _pvp, _pd, _vp, _vp, _d)


class VecOps(ctypes.Structure):
    """pn_vec_ops: NULL members select the library's HIP entry points."""
    _fields_ = [("rk_stage", RK_STAGE_FN), ("rk_combine_wrms", RK_COMBINE_WRMS_FN), ("adj_theta", ADJ_THETA_FN),
                ("adj_accum", ADJ_ACCUM_FN)]


# name -> (restype, argtypes); mirrors include/pnode_amd.h declaration by declaration
PROTOTYPES = {
    "pn_rk_attempt": (_i, [_vp, _i, _i64, _vp, _vp, _d, _d, _vp, _vp, _pvp, _vp, _i, _d, STAGE_CB, _vp, _i, _vp, _vp, _pvp]),
    "pn_rk_adjoint_step": (_i, [_vp, _i, _i64, _vp, _vp, _d, _d, _vp, _vp, _vp, VJP_CB, _vp, _vp]),
    "pn_last_error": (_cp, []),
    "pn_abi_version": (_i, []),
    "pn_tableau_get": (_i, [_cp, ctypes.POINTER(Tableau)]),
    "pn_method_to_rk_type": (_cp, [_cp]),
    "pn_rk_stage": (_i, [_vp, _i, _i64, _vp, _vp, _i, _pvp, _pd]),
    "pn_rk_combine_wrms": (_i, [_vp, _i, _i64, _vp, _vp, _i, _pvp, _pd, _pd, _d, _d, _vp, _vp]),
    "pn_wrms_work_bytes": (_i64, [_i64]),
    "pn_wrms_partials": (_i64, [_i64]),
    "pn_stream_wait_wrms": (_i, [_vp, _vp, _i64, _pd]),
    "pn_pinned_scalar": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "pn_pinned_free": (_i, [_vp]),
    "pn_stream_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "pn_stream_destroy": (_i, [_vp]),
    "pn_pinned_block": (_i, [_i64, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "pn_stream_wait_scalar": (_i, [_vp, _vp, _pd]),
    "pn_adj_theta": (_i, [_vp, _i, _i64, _vp, _vp, _d, _i, _pvp, _pd]),
    "pn_adj_accum": (_i, [_vp, _i, _i64, _vp, _vp, _i, _pvp, _pd, _vp, _vp, _d]),
    "pn_param_accum": (_i, [_vp, _i, _vp, _d, _i, _pvp, _pi64, _pi64]),
    "pn_param_accum_multi": (_i, [_vp, _i, _vp, _i, _pd, _i, _pvp, _pi64, _pi64]),
    "pn_lincomb": (_i, [_vp, _i, _i64, _vp, _i, _pvp, _pd]),
    "pn_dots": (_i, [_vp, _i, _i64, _vp, _i, _pvp, _vp, _vp]),
    "pn_dots_work_bytes": (_i64, [_i64]),
    "pn_stream_wait_scalars": (_i, [_vp, _vp, _i, _pd]),
    "pn_linear_wgrad_supported": (_i, [_i, _i64, _i64, _i64]),
    "pn_linear_wgrad_work_bytes": (_i64, [_i, _i64, _i64, _pi64]),
    "pn_linear_wgrad": (_i, [_vp, _i, _i64, _i64, _i64, _vp, _vp, _d, _vp, _vp]),
    "pn_linear_wgrad_group": (_i, [_vp, _i, _i64, _i, ctypes.POINTER(WgradPair), _i]),
    "pn_linear_wgrad_finish": (_i, [_vp, _i, _i64, _i64, _vp, _vp, _vp, _vp]),
    "pn_colsum_accum": (_i, [_vp, _i, _i64, _i64, _vp, _vp, _d, _vp]),
    "pn_colsum_accum_multi": (_i, [_vp, _i, _i, _pi64, _pi64, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _pd, _vp]),
    "pn_colsum_work_bytes": (_i64, [_i, _pi64, _pi64]),
    "pn_copy": (_i, [_vp, _i, _i64, _vp, _vp]),
    "pn_zero": (_i, [_vp, _i, _i64, _vp]),
    "pn_prof_enable": (_i, [_i]),
    "pn_prof_is_enabled": (_i, []),
    "pn_tune_set": (_i, [_cp]),
    "pn_prof_collect": (_i, [_i, _pi64, _pd, _pd]),
    "pn_kernel_name": (_cp, [_i]),
    "pn_ts_create": (_vp, []),
    "pn_ts_destroy": (None, [_vp]),
    "pn_ts_set_rk_type": (_i, [_vp, _cp]),
    "pn_ts_get_tableau": (_i, [_vp, ctypes.POINTER(Tableau)]),
    "pn_ts_set_option": (_i, [_vp, _cp, _cp]),
    "pn_ts_is_adaptive": (_i, [_vp]),
    "pn_ts_set_scheme": (_i, [_vp, _i, _i]),
    "pn_ts_get_tolerances": (_i, [_vp, _pd, _pd]),
    "pn_ts_begin": (_i, [_vp, _d, _d, _i, _pd]),
    "pn_ts_attempt": (_i, [_vp, _pd, _pd]),
    "pn_ts_judge": (_i, [_vp, _d, _pi, _pi, _pi]),
    "pn_ts_count_fixed_steps": (_i64, [_vp]),
    "pn_ts_override_next_dt": (_i, [_vp, _d]),
    "pn_ts_steps": (_i64, [_vp]),
    "pn_ts_rejections": (_i64, [_vp]),
    "pn_ts_time": (_d, [_vp]),
    "pn_ts_step_log": (_i, [_vp, _i64, _pd, _pd]),
    "pn_gmres_create": (_vp, [_i]),
    "pn_gmres_destroy": (None, [_vp]),
    "pn_gmres_begin": (_i, [_vp, _d]),
    "pn_gmres_column": (_i, [_vp, _i, _pd, _pd]),
    "pn_gmres_solve": (_i, [_vp, _i, _pd]),
    "pn_krylov_state_doubles": (_i64, [_i64, _i]),
    "pn_krylov_products_offset": (_i, [_i]),
    "pn_krylov_begin": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _i64, _vp, _d, _d, _i64, _i, _i]),
    "pn_krylov_step": (_i, [_vp, _i, _i64, _i, _vp, _vp, _i, _vp, _vp, _i64, _vp, _i]),
    "pn_krylov_close": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _i64]),
    "pn_traj_create": (_vp, []),
    "pn_traj_destroy": (None, [_vp]),
    "pn_traj_begin": (_i, [_vp, _i, _i64]),
    "pn_traj_set_carry": (_i, [_vp, _i]),
    "pn_traj_set_total": (_i, [_vp, _i64]),
    "pn_traj_fwd_slot": (_i64, [_vp, _i64]),
    "pn_traj_rev_plan": (_i, [_vp, _i64, _pi64, _pi64, _pi, _pi64, _pi64, _i]),
    "pn_traj_rev_done": (_i, [_vp, _i64]),
    "pn_traj_slots_in_use": (_i64, [_vp]),
    "pn_traj_dp_builds": (_i64, []),
    "pn_traj_high_water": (_i64, [_vp]),
    "pn_spill_create": (_vp, [_cp, _i64, _i, _i, _i]),
    "pn_spill_destroy": (None, [_vp]),
    "pn_spill_put": (_i, [_vp, _vp, _i64, _vp]),
    "pn_spill_prefetch": (_i, [_vp, _i64]),
    "pn_spill_get": (_i, [_vp, _vp, _i64, _vp]),
    "pn_spill_drop": (_i, [_vp, _i64]),
    "pn_spill_stats": (_i, [_vp, _pi64, _pi64, _pi64, _pi64]),
}

_lib = None


def load():
    """Load the shared library and bind every declared entry point (fails loudly)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pnode_amd: %s is missing -- the HIP library is the product and there is no fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950)." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.pn_abi_version() != PN_ABI_VERSION:
        raise ImportError("pnode_amd: ABI version mismatch (library %d, binding %d)" % (lib.pn_abi_version(), PN_ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc:
        raise PnError(load().pn_last_error().decode())


def dtype_code(dtype):
    if dtype == torch.float32:
        return PN_F32
    if dtype == torch.float64:
        return PN_F64
    raise TypeError("pnode_amd supports float32 and float64 states, got %s" % dtype)


def get_tableau(rk_type):
    t = Tableau()
    check(load().pn_tableau_get(rk_type.encode(), ctypes.byref(t)))
    return t
