"""GPU parity of the OPT-IN split-precision SDF query (csrc/query_x3.hip: fp32 operands split exactly into three bf16 planes, six
partial products on the bf16 matrix pipes, fp32 accumulation) -- held to the SAME budgets as the fp32 query kernel
(tests/test_gpu_query.py): 1e-5 absolute against the fp64 oracle and the reference's fp64 sdf vectors."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle_util import load_case
from test_gpu_query import _setup

pytestmark = pytest.mark.gpu


def _query_x3(lib, _lib, weff, use_deform, ld_out=0, ray_done=None, out=None, **kw):
    from endosurf_amd._lib import es_points
    M = kw["M"]
    px3 = torch.zeros(int(lib.es_packed_x3_bytes()), dtype=torch.uint8, device="cuda")
    _lib.check(lib.es_pack_x3(_lib.ptr(weff), _lib.ptr(px3), int(use_deform), _lib.stream_ptr()), "es_pack_x3")
    if out is None:
        out = torch.full((M,), float("nan"), device="cuda")
    pts = es_points()
    for k in ("x", "t", "dirs", "rays", "z"):
        setattr(pts, k, _lib.ptr(kw[k]) if kw.get(k) is not None else None)
    pts.mode, pts.t_scalar, pts.n_per_ray, pts.ldz, pts.M = kw.get("mode", 0), kw.get("t_scalar", 0), kw.get("n_per_ray", 1), kw.get("ldz", 1), M
    _lib.check(lib.es_query_sdf_x3(C.byref(pts), _lib.ptr(px3), _lib.ptr(weff), _lib.ptr(out), int(ld_out),
                                   _lib.ptr(ray_done) if ray_done is not None else None, int(use_deform), _lib.stream_ptr()), "es_query_sdf_x3")
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("mode,use_deform", [("init", True), ("trained", True), ("trained", False)])
@pytest.mark.parametrize("M", [1, 64, 1000, 20000])
def test_query_sdf_x3_points(mode, use_deform, M):
    lib, _lib, state, flat, weff, packed, net = _setup(21, mode, use_deform)
    rng = np.random.default_rng(5 + M)
    x = torch.from_numpy(rng.uniform(-0.8, 0.8, size=(M, 3)).astype(np.float32))
    t = torch.from_numpy(rng.uniform(size=(M,)).astype(np.float32))
    got = _query_x3(lib, _lib, weff, use_deform, x=x.cuda(), t=t.cuda(), M=M)
    with torch.no_grad():
        ref = net.sdf_observed(x.double(), t.double()[:, None])[:, 0]
    err = (got.double() - ref).abs().max().item()
    assert err < 1e-5, err
    # and it is as close to fp64 as the fp32 kernel is (split precision is not reduced precision)
    from test_gpu_query import _query
    got32 = _query(lib, _lib, packed, weff, use_deform, x=x.cuda(), t=t.cuda(), M=M)
    err32 = (got32.double() - ref).abs().max().item()
    assert err < 3 * err32 + 2e-6, (err, err32)


def test_query_sdf_x3_ray_samples_golden():
    c = load_case("trained_deform")
    lib, _lib, state, flat, weff, packed, net = _setup(int(c["meta/seed"]), "trained", True)
    rays = torch.from_numpy(c["rays"]).cuda()
    z = torch.from_numpy(c["z_trace/0"]).cuda().contiguous()
    N, n = z.shape
    got = _query_x3(lib, _lib, weff, True, rays=rays, z=z, mode=1, n_per_ray=n, ldz=n, M=N * n).reshape(N, n)
    assert np.max(np.abs(got.numpy() - c["sdf_trace64/0"])) < 1e-5
    assert np.max(np.abs(got.numpy() - c["sdf_trace/0"])) < 2e-5
    # strided output + ray_done skipping (the block-wise ray-marching form)
    out = torch.zeros(N, 2 * n, device="cuda")
    done = torch.zeros(N, dtype=torch.int32, device="cuda")
    done[: (N // 2) // 2 * 2] = 1                      # rays of whole 64-point tiles (2 rays x 32 samples) are skipped
    got2 = _query_x3(lib, _lib, weff, True, ld_out=2 * n, ray_done=done, out=out, rays=rays, z=z, mode=1, n_per_ray=n, ldz=n, M=N * n)
    k = int(done.sum())
    # (the flat small-batch call above runs the 32-point tiles, the strided one the 128-point blocks: same arithmetic, other summation order)
    assert torch.all(got2[:k] == 0) and np.max(np.abs(got2[k:, :n].numpy() - got.numpy()[k:])) < 2e-6 and torch.all(got2[:, n:] == 0)


def test_split_precision_training_step_matches_fp32():
    """engine.split_precision routes the coarse-sample and ray-marching queries through the x3 kernel: sample depths, ray-marching
    depths and the loss agree with the fp32 path within the budgets the fp32 path itself is held to."""
    from gpu_util import renderer_for
    from endosurf_amd.trainer import SyntheticScene, compute_loss_fused
    b = SyntheticScene("cuda", seed=8).batch(1024)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)                       # fixed draws: the number of samples that jump a bin depends on the jitter
    u, un = torch.rand(1024, 1, device="cuda", generator=gen), torch.rand(1024, 3, device="cuda", generator=gen)
    res = []
    for split in (False, True):
        r = renderer_for(24, "trained", True)
        r.engine.split_precision = split
        r.engine.march_block = 0
        with torch.no_grad():
            z = r.sample_z(b["rays"], 1, u_perturb=u)
            d_i = r.ray_marching(b["rays"])
        total, terms, _ = compute_loss_fused(r, b, 1, u_perturb=u, u_neigh=un)
        res.append((z, d_i, float(total)))
    (z0, d0, l0), (z1, d1, l1) = res
    # inverse-CDF sampling and the secant iteration amplify rounding differences of the SDF (the reference's own fp32-vs-fp64 sample
    # depths differ by up to ~1e-2 at isolated samples, tests/golden z_trace vs z_trace64; a sample in a flat stretch of the cdf can move by
    # a whole coarse bin, (far - near) / 32 ~ 0.06): quantile + a handful of outliers + bin-width bound, like test_gpu_rays.py
    dz = (z0 - z1).abs().flatten()
    q99, n_out, dmax = float(torch.quantile(dz, 0.99)), int((dz > 1e-3).sum()), float(dz.max())
    assert q99 < 2e-5 and n_out <= 32 and dmax < 0.1, (q99, n_out, dmax)      # of 65 536 sample depths
    fin = torch.isfinite(d0) & torch.isfinite(d1)
    assert torch.equal(torch.isfinite(d0), torch.isfinite(d1)) and float((d0[fin] - d1[fin]).abs().max()) < 2e-4
    assert abs(l0 - l1) < 1e-3 * max(1.0, abs(l0))
