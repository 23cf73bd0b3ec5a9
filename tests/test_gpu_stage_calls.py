"""The whole-stage C entry points (es_sample_z, es_render_forward, es_render_backward) against the drop-in renderer, which issues
the same launches one by one from Python: forward results must be identical, gradients equal up to the order of fp32 atomics."""
import ctypes as C

import pytest
import torch

from gpu_util import renderer_for

pytestmark = pytest.mark.gpu


def _setup(use_deform, N=96, seed=5):
    from endosurf_amd import _lib
    r = renderer_for(31, "trained", use_deform)
    from endosurf_amd.trainer import SyntheticScene
    rays = SyntheticScene("cuda", seed=seed).batch(N)["rays"].contiguous()
    weff, packed = r._weights()
    return r, _lib, rays, weff, packed


@pytest.mark.parametrize("use_deform", [True, False])
@pytest.mark.parametrize("perturb", [False, True])
def test_sample_z_matches_renderer(use_deform, perturb):
    r, _lib, rays, weff, packed = _setup(use_deform)
    lib, N = r.engine.lib, rays.shape[0]
    u = torch.rand(N, 1, device="cuda") if perturb else None
    with torch.no_grad():
        z_ref = r.sample_z(rays, iter_step=1, perturb_overwrite=perturb, u_perturb=u)
    S = r.n_samples + r.n_importance
    assert z_ref.shape == (N, S)
    z = torch.full((N, S), float("nan"), device="cuda")
    scratch = torch.empty(int(lib.es_sample_scratch_floats(N, r.n_samples, r.n_importance, r.up_sample_steps)), device="cuda")
    _lib.check(lib.es_sample_z(_lib.ptr(rays), _lib.ptr(u) if u is not None else None, N, r.n_samples, r.n_importance, r.up_sample_steps, 1,
                               _lib.ptr(packed.detach()), _lib.ptr(weff.detach()), int(use_deform), _lib.ptr(z), _lib.ptr(scratch),
                               _lib.stream_ptr()), "es_sample_z")
    torch.cuda.synchronize()
    assert torch.equal(z, z_ref)
    # coarse samples only
    z2 = torch.full((N, r.n_samples), float("nan"), device="cuda")
    _lib.check(lib.es_sample_z(_lib.ptr(rays), None, N, r.n_samples, r.n_importance, r.up_sample_steps, 0, _lib.ptr(packed.detach()),
                               _lib.ptr(weff.detach()), int(use_deform), _lib.ptr(z2), None, _lib.stream_ptr()), "es_sample_z")
    with torch.no_grad():
        r.n_importance, keep = 0, r.n_importance
        z2_ref = r.sample_z(rays, iter_step=1, perturb_overwrite=False)
        r.n_importance = keep
    assert torch.equal(z2, z2_ref)


@pytest.mark.parametrize("use_deform", [True, False])
def test_render_forward_backward_match_renderer(use_deform):
    r, _lib, rays, weff, packed = _setup(use_deform, N=64)
    lib, eng, N = r.engine.lib, r.engine, rays.shape[0]
    with torch.no_grad():
        z = r.sample_z(rays, iter_step=1, perturb_overwrite=False)
    S = z.shape[1]
    sample_dist, cos_anneal = 2.0 / r.n_samples, r.get_cos_anneal_ratio(1)
    # reference path: the drop-in's render_core + autograd
    r.zero_grad()
    ret = r.render_core(rays[:, :3], rays[:, 3:6], rays[:, 8], z, sample_dist, cos_anneal_ratio=cos_anneal, _rays=rays)
    g_color = torch.randn(N, 3, device="cuda")
    g_depth = torch.randn(N, 1, device="cuda")
    g_eik = torch.tensor(0.3, device="cuda")
    ((ret["color_map"] * g_color).sum() + (ret["depth_map"] * g_depth).sum() + ret["gradient_o_error"] * g_eik).backward()
    ref_grad = r.model._flat_grad.clone()
    ref_dvar = r.model.deviation_network.variance.grad.clone()

    # C path
    flags = (_lib.PF_DEFORM if use_deform else 0) | _lib.PF_SAVE
    a = _lib.es_render_args()
    var = r.model.deviation_network.variance.detach().reshape(1).contiguous()
    out = dict(color=eng.empty(N, 3), depth=eng.empty(N, 1), weights=eng.empty(N, S), cdf=eng.empty(N, S), weight_max=eng.empty(N, 1),
               eik_acc=eng.zeros(2), wmax_idx=eng.empty(N, dtype=torch.int32))
    ws = eng.empty(int(lib.es_point_workspace_floats(N * S, flags | _lib.PF_COLOR)))
    scratch = eng.empty(int(lib.es_render_scratch_floats(N, S)))
    a.c.rays, a.c.z, a.c.ldz, a.c.variance = _lib.ptr(rays), _lib.ptr(z), S, _lib.ptr(var)
    a.c.N, a.c.S, a.c.sample_dist, a.c.cos_anneal = N, S, sample_dist, cos_anneal
    for k, v in out.items():
        setattr(a.c, k, _lib.ptr(v))
    a.ws, a.scratch, a.flags = _lib.ptr(ws), _lib.ptr(scratch), flags
    wd, pd = weff.detach(), packed.detach()
    _lib.check(lib.es_render_forward(C.byref(a), _lib.ptr(pd), _lib.ptr(wd), _lib.stream_ptr()), "es_render_forward")
    torch.cuda.synchronize()
    assert torch.equal(out["color"], ret["color_map"].detach())
    assert torch.equal(out["depth"], ret["depth_map"].detach())
    assert torch.equal(out["weights"], ret["weights"].detach())
    assert torch.equal(out["cdf"], ret["cdf"].detach())
    eik_den = (out["eik_acc"][1] + 1e-6).reshape(1)
    eik = (out["eik_acc"][0] / eik_den[0]).item()                  # the eikonal sums are fp32 atomics: equal up to their order
    assert abs(eik - ret["gradient_o_error"].item()) <= 2e-6 * abs(eik)

    d_invs = eng.zeros(1)
    dweff = eng.zeros(eng.n_weff)
    a.c.g_color, a.c.g_depth, a.c.g_eik, a.c.eik_den = _lib.ptr(g_color), _lib.ptr(g_depth.view(-1)), _lib.ptr(g_eik.reshape(1)), _lib.ptr(eik_den)
    a.c.d_invs_acc = _lib.ptr(d_invs)
    _lib.check(lib.es_render_backward(C.byref(a), _lib.ptr(pd), _lib.ptr(wd), _lib.ptr(dweff), _lib.stream_ptr()), "es_render_backward")
    dflat = eng.weightnorm_backward(r.model._flat, dweff, use_deform)
    dvar = eng.variance_terms(var, d_invs_acc=d_invs)
    torch.cuda.synchronize()
    scale = ref_grad.abs().max().item()
    assert (dflat - ref_grad).abs().max().item() <= 2e-5 * scale
    assert abs(dvar.item() - ref_dvar.item()) <= 1e-5 * max(abs(ref_dvar.item()), 1e-6)


@pytest.mark.parametrize("use_deform", [True, False])
def test_render_forward_split_precision_stage_call(use_deform):
    """es_render_forward with es_render_args.packed_x3 + ES_PF_X3 (no ES_PF_SAVE): the opt-in split-precision inference chain behind
    the whole-stage C call agrees with the fp32 stage call (same inputs, same workspace layout)."""
    r, _lib, rays, weff, packed = _setup(use_deform, N=96)
    lib, eng, N = r.engine.lib, r.engine, rays.shape[0]
    with torch.no_grad():
        z = r.sample_z(rays, iter_step=1, perturb_overwrite=False)
    S = z.shape[1]
    var = r.model.deviation_network.variance.detach().reshape(1).contiguous()
    wd, pd = weff.detach(), packed.detach()
    px3 = torch.zeros(int(lib.es_packed_x3_bytes()), dtype=torch.uint8, device="cuda")
    _lib.check(lib.es_pack_x3(_lib.ptr(wd), _lib.ptr(px3), int(use_deform), _lib.stream_ptr()), "es_pack_x3")
    res = []
    for split in (False, True):
        flags = (_lib.PF_DEFORM if use_deform else 0) | (_lib.PF_X3 if split else 0)
        a = _lib.es_render_args()
        out = dict(color=eng.empty(N, 3), depth=eng.empty(N, 1), weights=eng.empty(N, S), cdf=eng.empty(N, S), weight_max=eng.empty(N, 1),
                   eik_acc=eng.zeros(2), wmax_idx=eng.empty(N, dtype=torch.int32))
        ws = eng.empty(int(lib.es_point_workspace_floats(N * S, flags | _lib.PF_COLOR)))
        scratch = eng.empty(int(lib.es_render_scratch_floats(N, S)))
        a.c.rays, a.c.z, a.c.ldz, a.c.variance = _lib.ptr(rays), _lib.ptr(z), S, _lib.ptr(var)
        a.c.N, a.c.S, a.c.sample_dist, a.c.cos_anneal = N, S, 2.0 / r.n_samples, r.get_cos_anneal_ratio(1)
        for k, v in out.items():
            setattr(a.c, k, _lib.ptr(v))
        a.ws, a.scratch, a.flags = _lib.ptr(ws), _lib.ptr(scratch), flags
        a.packed_x3 = _lib.ptr(px3) if split else None
        _lib.check(lib.es_render_forward(C.byref(a), _lib.ptr(pd), _lib.ptr(wd), _lib.stream_ptr()), "es_render_forward")
        torch.cuda.synchronize()
        res.append(out)
    f32, x3 = res
    for k, tol_q, tol_max in (("color", 2e-4, 5e-2), ("depth", 2e-4, 5e-2), ("weights", 2e-4, 5e-2)):
        d = (f32[k] - x3[k]).abs().flatten()
        assert float(torch.quantile(d, 0.99)) < tol_q and float(d.max()) < tol_max, (k, float(torch.quantile(d, 0.99)), float(d.max()))
    assert not torch.equal(f32["color"], x3["color"]), "the split-precision path must have run (results differ in the last bits)"


@pytest.mark.parametrize("use_deform", [True, False])
@pytest.mark.parametrize("N,block", [(96, 32), (512, 32), (512, 0)])
def test_ray_marching_matches_renderer(use_deform, N, block):
    r, _lib, rays, weff, packed = _setup(use_deform, N=N, seed=9)
    lib = r.engine.lib
    r.engine.split_precision = False                   # es_ray_marching is the fp32 stage: bit-equality holds against the fp32 orchestration
    with torch.no_grad():
        d_ref = r.ray_marching(rays)                    # engine path: block-wise only for N * 32 >= 16384, else one launch
    d = torch.full((N, 1), float("nan"), device="cuda")
    scratch = torch.empty(int(lib.es_march_scratch_floats(N, 128)), device="cuda")
    _lib.check(lib.es_ray_marching(_lib.ptr(rays), N, 128, 8, 0.0, block, _lib.ptr(packed.detach()), _lib.ptr(weff.detach()), int(use_deform),
                                   _lib.ptr(d), _lib.ptr(scratch), _lib.stream_ptr()), "es_ray_marching")
    torch.cuda.synchronize()
    assert torch.equal(d, d_ref)
