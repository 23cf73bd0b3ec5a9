"""es_train_loss (one launch) vs. the same loss written with torch ops (the reference's compute_loss arithmetic,
trainer_endosurf.py:133-162, endosurf.py:306-315, :337-339): values and gradients w.r.t. every renderer output."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_loss(color_map, depth_map, eik, a_sdf, a_go, rays, eod_pts, color_gt, depth_gt, mask, cmask, valid_sn, w):
    N = rays.shape[0]
    color_loss = ((color_map - color_gt) * cmask).abs().sum() / (cmask.sum() + 1e-10)
    cos = (rays[:, 3:6] * a_go[:N]).sum(-1, keepdim=True)
    inside = (torch.linalg.norm(eod_pts, dim=-1, keepdim=True) < 1.0).float() * mask
    den = inside.sum() + 1e-6
    sdf_loss = (inside * a_sdf[:N]).abs().sum() / den
    angle_loss = torch.relu(cos).abs().sum() / den
    depth_loss = ((depth_map - depth_gt) * inside * mask).abs().sum() / ((inside * mask).sum() + 1e-10)
    g = a_go[N:]
    normal = g / (torch.linalg.norm(g, dim=-1, keepdim=True) + 1e-10)
    diff = (normal[:N] - normal[N:]).abs() * valid_sn[:, None].float()
    sn = diff.sum() / torch.clamp(valid_sn.sum() * 3, min=1).float()
    terms = dict(color=color_loss, depth=depth_loss, sdf=sdf_loss, angle=angle_loss, eikonal=eik, surf_neig=sn)
    return sum(w[k] * terms[k] for k in terms), terms


@pytest.mark.parametrize("N,p_valid", [(1024, 0.7), (37, 0.5), (5000, 0.0)])
def test_loss_kernel_matches_torch(N, p_valid):
    from endosurf_amd.engine import Engine
    from endosurf_amd.trainer import LOSS_WEIGHTS, _LossFn
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(N)
    R = lambda *s: torch.randn(*s, generator=g).to(dev)
    U = lambda *s: torch.rand(*s, generator=g).to(dev)
    rays = R(N, 9)
    outs = [U(N, 3), 1.2 + 0.3 * R(N, 1), U(1).reshape(()) * 0.1, 0.1 * R(3 * N, 1), R(3 * N, 3)]
    eod_pts = 0.8 * R(N, 3)                                    # a good share outside the unit sphere
    color_gt, depth_gt = U(N, 3), 1.2 + 0.3 * R(N, 1)
    mask, cmask = (U(N, 1) < 0.8).float(), (U(N, 1) < 0.6).float()
    valid_sn = U(N) < p_valid
    w = dict(LOSS_WEIGHTS)
    w.update(color=1.0, depth=0.7, sdf=0.3, angle=0.2, eikonal=0.1, surf_neig=0.05)
    a = [t.clone().requires_grad_(True) for t in outs]
    b = [t.clone().requires_grad_(True) for t in outs]
    tot_t, terms_t = _torch_loss(*a, rays, eod_pts, color_gt, depth_gt, mask, cmask, valid_sn, w)
    (2.5 * tot_t).backward()
    tot_k, t = _LossFn.apply(*b, Engine(dev), rays, eod_pts, color_gt, depth_gt, mask, cmask, valid_sn, w)
    (2.5 * tot_k).backward()
    torch.cuda.synchronize()
    assert abs(float(tot_t) - float(tot_k)) <= 2e-6 * max(1.0, abs(float(tot_t)))
    for i, k in enumerate(("color", "depth", "sdf", "angle", "eikonal", "surf_neig")):
        assert abs(float(terms_t[k]) - float(t[i])) <= 2e-6 * max(1.0, abs(float(terms_t[k]))), k
    assert float(t[7]) == float(valid_sn.sum())
    for x, y, name in zip(a, b, ("color_map", "depth_map", "eik", "aux_sdf", "aux_go")):
        gx = x.grad if x.grad is not None else torch.zeros_like(x)
        assert y.grad is not None and y.grad.shape == x.shape, name
        err = float((gx - y.grad).abs().max())
        assert err <= 2e-6 * max(1e-3, float(gx.abs().max())) + 1e-9, (name, err)
