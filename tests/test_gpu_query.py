"""GPU parity: weight-norm/pack and the fused SDF query kernel (K2) against the fp64 oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

import weightgen
from oracle import endosurf_oracle as O
from oracle_util import RENDER_CFG, T, load_case, oracle_for

pytestmark = pytest.mark.gpu


def _setup(seed, mode, use_deform):
    from endosurf_amd import _lib, params
    lib = _lib.load()
    _lib.check(lib.es_init(), "es_init")
    state = weightgen.make_state(seed, mode, use_deform)
    flat = torch.from_numpy(params.flatten_state(state)).cuda()
    weff = torch.zeros(lib.es_weff_floats(), device="cuda")
    packed = torch.zeros(lib.es_packed_floats(), device="cuda")
    _lib.check(lib.es_weightnorm_pack(_lib.ptr(flat), _lib.ptr(weff), _lib.ptr(packed), int(use_deform), _lib.stream_ptr()))
    torch.cuda.synchronize()
    net = O.OracleNet({k: torch.tensor(v, dtype=torch.float64) for k, v in state.items()}, use_deform)
    return lib, _lib, state, flat, weff, packed, net


@pytest.mark.parametrize("use_deform", [True, False])
def test_weightnorm_pack_matches_oracle(use_deform):
    from endosurf_amd import params
    lib, _lib, state, flat, weff, packed, net = _setup(11, "trained", use_deform)
    w = weff.cpu().numpy()
    for (ni, l), (w_off, b_off, N, K) in params.weff_layout().items():
        name = params.NET_NAMES[ni]
        if name == "deform_network" and not use_deform:
            continue
        W, b = net._wb(name, l)
        assert np.max(np.abs(w[w_off:w_off + N * K].reshape(N, K) - W.numpy())) < 2e-6, (name, l)
        assert np.array_equal(w[b_off:b_off + N], b.numpy().astype(np.float32)), (name, l)
    assert torch.isfinite(packed).all()


def _query(lib, _lib, packed, weff, use_deform, **kw):
    from endosurf_amd._lib import es_points
    M = kw["M"]
    out = torch.full((M,), float("nan"), device="cuda")
    pts = es_points()
    for k in ("x", "t", "dirs", "rays", "z"):
        setattr(pts, k, _lib.ptr(kw[k]) if kw.get(k) is not None else None)
    pts.mode, pts.t_scalar, pts.n_per_ray, pts.ldz, pts.M = kw.get("mode", 0), kw.get("t_scalar", 0), kw.get("n_per_ray", 1), kw.get("ldz", 1), M
    if kw.get("tile_points"):
        _lib.check(lib.es_query_sdf_tiles(C.byref(pts), _lib.ptr(packed), _lib.ptr(weff), _lib.ptr(out), int(use_deform), int(kw["tile_points"]),
                                          _lib.stream_ptr()), "es_query_sdf_tiles")
    else:
        _lib.check(lib.es_query_sdf(C.byref(pts), _lib.ptr(packed), _lib.ptr(weff), _lib.ptr(out), int(use_deform), _lib.stream_ptr()), "es_query_sdf")
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("mode,use_deform", [("init", True), ("trained", True), ("trained", False)])
@pytest.mark.parametrize("M", [1, 64, 1000, 12000, 20031])      # 16-point tiles (<= 9 216) | 32-point tiles (<= 16 384) | 64-point tiles, ragged
def test_query_sdf_points(mode, use_deform, M):
    lib, _lib, state, flat, weff, packed, net = _setup(21, mode, use_deform)
    rng = np.random.default_rng(5 + M)
    x = torch.from_numpy(rng.uniform(-0.8, 0.8, size=(M, 3)).astype(np.float32))
    t = torch.from_numpy(rng.uniform(size=(M,)).astype(np.float32))
    got = _query(lib, _lib, packed, weff, use_deform, x=x.cuda(), t=t.cuda(), M=M)
    with torch.no_grad():
        ref = net.sdf_observed(x.double(), t.double()[:, None])[:, 0]
    err = (got.double() - ref).abs().max().item()
    # fp32 network evaluation noise of the reference itself is ~1e-6 on sdf (tests/test_oracle_golden.py)
    assert err < 1e-5, err
    # scalar-time variant
    got2 = _query(lib, _lib, packed, weff, use_deform, x=x.cuda(), t=t[:1].cuda().contiguous(), t_scalar=1, M=M)
    with torch.no_grad():
        ref2 = net.sdf_observed(x.double(), t.double()[:1].expand(M)[:, None])[:, 0]
    assert (got2.double() - ref2).abs().max().item() < 1e-5


def test_query_sdf_tiles_choice_of_tile_height():
    """es_query_sdf_tiles (ABI v6): the caller's tile height.  20 031 points on 16-, 32- and 64-point tiles: each within the fp32 budget of
    the fp64 oracle, within 2e-6 of each other (summation order), the 64-point choice bit-identical to es_query_sdf's own."""
    lib, _lib, state, flat, weff, packed, net = _setup(23, "trained", True)
    rng = np.random.default_rng(9)
    M = 20031
    x = torch.from_numpy(rng.uniform(-0.8, 0.8, size=(M, 3)).astype(np.float32))
    t = torch.from_numpy(rng.uniform(size=(M,)).astype(np.float32))
    auto = _query(lib, _lib, packed, weff, True, x=x.cuda(), t=t.cuda(), M=M)
    got = {tp: _query(lib, _lib, packed, weff, True, x=x.cuda(), t=t.cuda(), M=M, tile_points=tp) for tp in (16, 32, 64)}
    with torch.no_grad():
        ref = net.sdf_observed(x.double(), t.double()[:, None])[:, 0]
    for tp, v in got.items():
        assert (v.double() - ref).abs().max().item() < 1e-5, tp
        assert (v - auto).abs().max().item() < 2e-6, tp
    assert torch.equal(got[64], auto) and torch.equal(got[32], got[64])          # 32- and 64-point tiles: the same MFMA order per row
    assert not torch.equal(got[16], auto)                                         # (the 16-point kernel did run)


def test_query_sdf_ray_samples_golden():
    """Mode 1 (ray samples) against sdf values captured from the reference (coarse samples of the golden case)."""
    c = load_case("trained_deform")
    lib, _lib, state, flat, weff, packed, net = _setup(int(c["meta/seed"]), "trained", True)
    rays = torch.from_numpy(c["rays"]).cuda()
    z = torch.from_numpy(c["z_trace/0"]).cuda().contiguous()
    N, n = z.shape
    got = _query(lib, _lib, packed, weff, True, rays=rays, z=z, mode=1, n_per_ray=n, ldz=n, M=N * n).reshape(N, n)
    ref32 = c["sdf_trace/0"]
    ref64 = c["sdf_trace64/0"]
    assert np.max(np.abs(got.numpy() - ref64)) < 1e-5
    assert np.max(np.abs(got.numpy() - ref32)) < 2e-5
