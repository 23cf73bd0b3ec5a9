"""ctypes binding of libendosurf_hip.so (C ABI declared in include/endosurf_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol cannot be
resolved, importing/using the renderer raises.  Build it with ``python -m endosurf_amd.build``.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libendosurf_hip.so")

_c_float_p = C.c_void_p   # device pointers travel as integers (torch .data_ptr())


class es_points(C.Structure):
    _fields_ = [("x", C.c_void_p), ("t", C.c_void_p), ("dirs", C.c_void_p), ("rays", C.c_void_p), ("z", C.c_void_p),
                ("mode", C.c_int), ("t_scalar", C.c_int), ("n_per_ray", C.c_int), ("ldz", C.c_int), ("M", C.c_int), ("M_split", C.c_int)]


class es_composite_args(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("rays", "z")] + [("ldz", C.c_int)]
                + [(n, C.c_void_p) for n in ("sdf", "g_o", "rgb", "variance")]
                + [("N", C.c_int), ("S", C.c_int), ("sample_dist", C.c_float), ("cos_anneal", C.c_float)]
                + [(n, C.c_void_p) for n in ("color", "depth", "weights", "cdf", "weight_max", "eik_acc", "wmax_idx",
                                             "g_color", "g_depth", "g_weights", "g_cdf", "g_wmax", "g_gradients_o", "g_eik",
                                             "eik_den", "d_sdf", "d_go", "d_rgb", "d_invs_acc", "ray_part", "cos_anneal_dev",
                                             "go_copy", "g_aux_sdf", "g_aux_go")] + [("n_aux", C.c_int)])


class es_render_args(C.Structure):
    _fields_ = [("c", es_composite_args), ("ws", C.c_void_p), ("scratch", C.c_void_p), ("flags", C.c_int), ("wg_scratch", C.c_void_p), ("packed_x3", C.c_void_p)]


class es_loss_args(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("color_map", "depth_map", "eik", "aux_sdf", "aux_go", "rays", "eod_pts", "color_gt", "depth_gt",
                                            "mask", "cmask", "valid_sn")]
                + [("N", C.c_int)] + [(n, C.c_float) for n in ("w_color", "w_depth", "w_sdf", "w_angle", "w_eik", "w_sn")]
                + [(n, C.c_void_p) for n in ("terms", "g_color", "g_depth", "g_eik", "g_aux_sdf", "g_aux_go", "den_out", "den_global")]
                + [("world", C.c_float), ("total_out", C.c_void_p)])


_P = C.c_void_p
_I = C.c_int
_F = C.c_float

# name -> (restype, argtypes); every symbol declared in include/endosurf_hip.h must be listed here
# (tests/test_abi.py cross-checks the header against this table and against the built .so).
PROTOTYPES = {
    "es_abi_version": (C.c_int, []),
    "es_last_error": (C.c_char_p, []),
    "es_init": (C.c_int, []),
    "es_param_floats": (C.c_int64, []),
    "es_param_layout": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "es_param_variance_off": (C.c_int64, []),
    "es_weff_floats": (C.c_int64, []),
    "es_weff_layout": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "es_packed_floats": (C.c_int64, []),
    "es_weightnorm_pack": (C.c_int, [_c_float_p, _c_float_p, _c_float_p, C.c_int, C.c_void_p]),
    "es_weightnorm_backward": (C.c_int, [_c_float_p, _c_float_p, _c_float_p, C.c_int, C.c_void_p]),
    "es_query_sdf": (C.c_int, [C.POINTER(es_points), _c_float_p, _c_float_p, _c_float_p, C.c_int, C.c_void_p]),
    "es_query_sdf_tiles": (C.c_int, [C.POINTER(es_points), _c_float_p, _c_float_p, _c_float_p, C.c_int, C.c_int, C.c_void_p]),
    "es_packed_x3_bytes": (C.c_int64, []),
    "es_pack_x3": (_I, [_P, _P, _I, _P]),
    "es_query_sdf_x3": (_I, [C.POINTER(es_points), _P, _P, _P, _I, _P, _I, _P]),
    "es_ray_setup": (_I, [_P, _P, _I, _I, _F, _I, _P, _I, _P, _P, _P]),
    "es_upsample_step": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _F, _P, _P, _I, _P, _P]),
    "es_merge_sdf": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _P, _P]),
    "es_mid_z": (_I, [_P, _I, _I, _I, _F, _P, _P]),
    "es_composite_forward": (_I, [C.POINTER(es_composite_args), _P]),
    "es_composite_backward": (_I, [C.POINTER(es_composite_args), _P]),
    "es_march_find": (_I, [_P, _P, _I, _I, _F, _P, _P, _P, _P]),
    "es_secant_points": (_I, [_P, _P, _I, _P, _P, _P]),
    "es_secant_update": (_I, [_P, _I, _F, _P, _P, _P]),
    "es_march_finish": (_I, [_P, _P, _I, _P, _P]),
    "es_point_workspace_floats": (C.c_int64, [_I, _I]),
    "es_point_workspace_offset": (C.c_int64, [_I, _I, _I]),
    "es_point_forward": (_I, [C.POINTER(es_points), _P, _P, _P, _I, _I, _P]),
    "es_point_forward_rows": (_I, [C.POINTER(es_points), _P, _P, _P, _I, _I, _I, _I, _P]),
    "es_eod_points": (_I, [_P, _P, _P, _I, _P, _P, _P, _P]),
    "es_sn_points": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P, _P]),
    "es_eod_loss": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "es_eod_loss_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "es_sn_loss": (_I, [_P, _P, _I, _P, _P]),
    "es_sn_loss_backward": (_I, [_P, _P, _P, _P, _I, _P, _P]),
    "es_copy2": (_I, [_P, _P, C.c_longlong, _P, _P, C.c_longlong, _P]),
    "es_color_forward": (_I, [C.POINTER(es_points), _P, _P, _P, _P]),
    "es_point_vjp": (_I, [C.POINTER(es_points), _P, _P, _P, _I, _P]),
    "es_point_backward_stages": (_I, [C.POINTER(es_points), _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "es_weightnorm_backward_layers": (_I, [_P, _P, _P, _I, _I, _P]),
    "es_point_forward_x3": (_I, [C.POINTER(es_points), _P, _P, _P, _P, _I, _I, _P]),
    "es_gemm_atb": (_I, [_P, _P, _I, _P, _I, _P, _P]),
    "es_point_backward_x3": (_I, [C.POINTER(es_points), _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "es_point_backward": (_I, [C.POINTER(es_points), _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "es_wgrad_scratch_floats": (C.c_int64, []),
    "es_point_backward_det": (_I, [C.POINTER(es_points), _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "es_train_loss": (_I, [C.POINTER(es_loss_args), _P]),
    "es_query_sdf_rays": (_I, [C.POINTER(es_points), _P, _P, _P, _I, _P, _I, _P]),
    "es_march_progress": (_I, [_P, _I, _I, _I, C.c_float, _P, _P]),
    "es_variance_terms": (_I, [_P, _P, _P, _P, _P]),
    "es_sample_scratch_floats": (C.c_int64, [_I, _I, _I, _I]),
    "es_sample_z": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "es_march_scratch_floats": (C.c_int64, [_I, _I]),
    "es_ray_marching": (_I, [_P, _I, _I, _I, C.c_float, _I, _P, _P, _I, _P, _P, _P]),
    "es_render_scratch_floats": (C.c_int64, [_I, _I]),
    "es_render_forward": (_I, [C.POINTER(es_render_args), _P, _P, _P]),
    "es_render_backward": (_I, [C.POINTER(es_render_args), _P, _P, _P, _P]),
    "es_train_aux_points": (_I, [_P, _P, _P, _P, _P, C.c_float, _I, _P, _P, _P, _P]),
    "es_adam_step": (_I, [_P, _P, _P, _P, C.c_longlong, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, C.c_longlong, _P]),
    "es_train_schedule": (_I, [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_float, C.c_double, _P, _P]),
    "es_adam_step_dev": (_I, [_P, _P, _P, _P, C.c_longlong, C.c_float, C.c_float, C.c_float, _P, _P, C.c_longlong, _P]),
    "es_zero": (_I, [_P, C.c_longlong, _P]),
    "es_uniform": (_I, [_P, C.c_longlong, C.c_ulonglong, C.c_ulonglong, _P, _P]),
    "es_scale": (_I, [_P, _P, C.c_longlong, _P, _P]),
    "es_render_finish": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "es_timing_enable": (_I, [_I]),
    "es_timing_drain": (_I, [_I, C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "es_kernel_name": (C.c_char_p, [_I]),
}

ABI_VERSION = 8
QUERY_TILE_RACING = 32      # include/endosurf_hip.h ES_QUERY_TILE_RACING
PF_DEFORM, PF_COLOR, PF_SAVE, PF_X3 = 1, 2, 4, 8
WS_XC, WS_V, WS_SDF, WS_FEAT, WS_GC, WS_GO, WS_RGB = range(7)
BWD_CHAINS, BWD_WGRAD_DEFORM, BWD_WGRAD_SDF, BWD_WGRAD_COLOR = 1, 2, 4, 8          # es_point_backward_stages
WS_XCBAR, WS_CURV, WS_TBAR, WS_VBAR = 27, 34, 35, 23          # (include/endosurf_hip.h ES_WS_XCBAR / ES_WS_CURV / ES_WS_TBAR / ES_WS_VBAR)

_lib = None


class EndoSurfHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once). Raises if it has not been built — there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EndoSurfHipError(
            f"{LIB_PATH} not found: build the HIP extension first (python -m endosurf_amd.build). "
            "endosurf_amd has no CPU/PyTorch fallback for the renderer hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.es_abi_version() != ABI_VERSION:
        raise EndoSurfHipError("libendosurf_hip ABI version mismatch")
    _lib = lib
    return lib


calls = 0          # library calls checked so far (a measurement counter: bench.py reports library calls per step)


def check(status: int, what: str = ""):
    global calls
    calls += 1
    if status != 0:
        msg = load().es_last_error()
        raise EndoSurfHipError(f"{what or 'libendosurf_hip'} failed (status {status}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a contiguous fp32 torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "C ABI takes contiguous buffers"
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """torch's current HIP stream on ``device`` (default: the current device)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
