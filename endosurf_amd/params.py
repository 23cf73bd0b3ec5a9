"""Flat parameter buffer <-> reference state_dict tensors (layout owned by csrc/arch.h, queried through the C ABI)."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np

from . import _lib

NET_NAMES = ("deform_network", "sdf_network", "color_network")


def layout():
    """{key: (offset, shape)} for every reference state_dict key ("sdf_network.net.3.weight_v", ...)."""
    lib = _lib.load()
    out = {}
    for ni, net in enumerate(NET_NAMES):
        for l in range(9):
            b, g, v = C.c_int64(), C.c_int64(), C.c_int64()
            n, k = C.c_int(), C.c_int()
            _lib.check(lib.es_param_layout(ni, l, C.byref(b), C.byref(g), C.byref(v), C.byref(n), C.byref(k)))
            out[f"{net}.net.{l}.bias"] = (b.value, (n.value,))
            out[f"{net}.net.{l}.weight_g"] = (g.value, (n.value, 1))
            out[f"{net}.net.{l}.weight_v"] = (v.value, (n.value, k.value))
    out["deviation_network.variance"] = (lib.es_param_variance_off(), ())
    return out


def weff_layout():
    """{(net_idx, layer): (w_off, b_off, N, K)} of the effective-weight buffer."""
    lib = _lib.load()
    lay = layout()
    out = {}
    for ni, net in enumerate(NET_NAMES):
        for l in range(9):
            w, b = C.c_int64(), C.c_int64()
            _lib.check(lib.es_weff_layout(ni, l, C.byref(w), C.byref(b)))
            n, k = lay[f"{net}.net.{l}.weight_v"][1]
            out[(ni, l)] = (w.value, b.value, n, k)
    return out


def flatten_state(state: Dict[str, np.ndarray]) -> np.ndarray:
    """Reference-format state (numpy) -> flat fp32 vector. Missing deform keys are left zero (use_deform=False)."""
    lib = _lib.load()
    flat = np.zeros(lib.es_param_floats(), np.float32)
    for key, (off, shape) in layout().items():
        if key in state:
            a = np.asarray(state[key], np.float32).reshape(-1)
            assert a.size == int(np.prod(shape)) if shape else a.size == 1, key
            flat[off:off + a.size] = a
    return flat
