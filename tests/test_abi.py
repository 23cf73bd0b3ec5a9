"""CPU: the C-ABI library loads and exports every symbol include/endosurf_hip.h declares (no compute calls without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "endosurf_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(es_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from endosurf_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def test_header_and_prototypes_agree(lib):
    from endosurf_amd import _lib
    decl = declared_functions()
    assert decl, "no declarations parsed"
    assert sorted(_lib.PROTOTYPES) == decl


def test_every_declared_symbol_is_exported(lib):
    for name in declared_functions():
        assert getattr(lib, name) is not None, name
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(REPO, "endosurf_amd", "lib", "libendosurf_hip.so")],
                         capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (es_[a-z0-9_]+)", out))
    assert set(declared_functions()) <= exported


def test_layout_queries_match_reference_shapes(lib):
    from endosurf_amd import _lib, params
    import weightgen
    assert lib.es_abi_version() == _lib.ABI_VERSION
    assert lib.es_param_floats() == 1654951            # reference parameter count
    lay = params.layout()
    state = weightgen.make_state(1, "init", True)
    assert set(lay) == set(state)
    end = 0
    for key, (off, shape) in sorted(lay.items(), key=lambda kv: kv[1][0]):
        assert off == end, key
        assert tuple(shape) == tuple(state[key].shape), key
        n = 1
        for s in shape:
            n *= s
        end = off + n
    assert end == lib.es_param_floats()
    flat = params.flatten_state(state)
    off, shape = lay["sdf_network.net.4.weight_v"]
    assert (flat[off:off + 256 * 295].reshape(256, 295) == state["sdf_network.net.4.weight_v"]).all()


def test_struct_sizes_match_header(lib):
    from endosurf_amd import _lib
    assert C.sizeof(_lib.es_points) == 5 * 8 + 6 * 4          # 5 pointers, 6 ints
    assert C.sizeof(_lib.es_composite_args) % 8 == 0
    assert lib.es_point_workspace_floats(0, 7) == 0
    n = lib.es_point_workspace_floats(100, 7)
    assert n > 128 * 20000 and lib.es_point_workspace_offset(100, 7, 2) >= 128 * 6      # x_c (3) and v = J d (3) precede the sdf buffer
    assert lib.es_kernel_name(0) == b"k_query_sdf"


def test_bad_arguments_return_error_codes(lib):
    from endosurf_amd import _lib
    assert lib.es_param_layout(7, 0, None, None, None, None, None) == 1
    assert b"out of range" in lib.es_last_error()
    p = _lib.es_points()
    p.M, p.mode = 4, 0
    assert lib.es_query_sdf(C.byref(p), None, None, None, 1, None) == 1      # null x/t
    p.mode = 3
    assert lib.es_query_sdf(C.byref(p), None, None, None, 1, None) == 1
    # es_query_sdf_tiles (ABI v6): the tile height is 0 (by batch size), 16, 32 or 64 -- checked before anything is launched
    import numpy as np
    buf = np.zeros(64, np.float32)
    q = _lib.es_points()
    q.M, q.mode, q.t_scalar = 4, 0, 1
    q.x, q.t = buf.ctypes.data, buf.ctypes.data          # (host addresses: never dereferenced, the call fails on its argument check)
    assert lib.es_query_sdf_tiles(C.byref(q), buf.ctypes.data, buf.ctypes.data, buf.ctypes.data, 1, 24, None) == 1
    assert b"tile_points" in lib.es_last_error()
