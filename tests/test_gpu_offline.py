"""Offline helpers on the HIP path (renderonpts, renderondepth, extract_fields, extract_observation_geometry) against the
vectors captured from the reference (tests/golden/offline_*.npz).  Tolerance = 3x the reference's own fp32-vs-fp64 error."""
import numpy as np
import pytest
import torch

from gpu_util import renderer_for_case
from oracle_util import load_case

pytestmark = pytest.mark.gpu
OFFLINE = ["trained_deform", "trained_nodeform"]


def _q(a, b, q=0.99):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.quantile(np.abs(a.astype(np.float64) - np.asarray(b, np.float64)), q))


def _budget(g, key, floor, q=0.99):
    return 3 * _q(g[key], g[key.replace("/", "64/", 1)], q) + floor


@pytest.mark.parametrize("name", OFFLINE)
def test_renderonpts(name):
    c, g = load_case(name), load_case("offline_" + name)
    r = renderer_for_case(c)
    dev = "cuda"
    x, d, t = (torch.from_numpy(c[k]).to(dev) for k in ("pt/x", "pt/d", "pt/t"))
    color, normal = r.renderonpts(x, d, t, cpu=True)
    assert isinstance(normal, np.ndarray) and torch.is_tensor(color) and color.is_cuda
    assert _q(color, g["onpts64/color"]) <= _budget(g, "onpts/color", 2e-6)
    assert _q(normal, g["onpts64/normal"]) <= _budget(g, "onpts/normal", 2e-5)
    # shared-time form, leading shape kept, chunked launches
    color, normal = r.renderonpts(x.reshape(8, -1, 3), d.reshape(8, -1, 3), torch.tensor([0.37]), net_chunk=50, cpu=False)
    assert tuple(color.shape) == (8, 24, 3) and tuple(normal.shape) == (8, 24, 3) and normal.is_cuda
    assert _q(color, g["onpts164/color"]) <= _budget(g, "onpts1/color", 2e-6)
    assert _q(normal, g["onpts164/normal"]) <= _budget(g, "onpts1/normal", 2e-5)


@pytest.mark.parametrize("name", OFFLINE)
def test_renderondepth(name):
    c, g = load_case(name), load_case("offline_" + name)
    r = renderer_for_case(c)
    rays = torch.from_numpy(c["rays"]).cuda()
    depth = torch.from_numpy(g["ondepth/depth_in"]).cuda()
    col, grad, d_out = r.renderondepth(rays, depth)
    valid = (g["ondepth/depth_in"][:, 0] > 0) & np.isfinite(g["ondepth/depth_in"][:, 0])
    assert _q(col, g["ondepth64/color"]) <= _budget(g, "ondepth/color", 2e-6)
    assert _q(grad, g["ondepth64/gradients"]) <= _budget(g, "ondepth/gradients", 2e-5)
    assert float(col[torch.from_numpy(~valid).cuda()].abs().max()) == 0 and float(grad[torch.from_numpy(~valid).cuda()].abs().max()) == 0
    assert _q(d_out, g["ondepth64/d_out"], 1.0) < 2e-6 and bool(torch.isfinite(d_out).all())
    col, grad, d_out = r.renderondepth(rays, torch.zeros_like(depth))
    assert float(col.abs().max()) == 0 and float(grad.abs().max()) == 0 and float(d_out.abs().max()) == 0


@pytest.mark.parametrize("name", OFFLINE)
def test_extract_fields_and_geometry(name):
    c, g = load_case(name), load_case("offline_" + name)
    r = renderer_for_case(c)
    res = int(g["meta/res"])
    bmin, bmax, t = torch.from_numpy(g["fields/bmin"]), torch.from_numpy(g["fields/bmax"]), torch.tensor([float(g["fields/t"])])
    u = r.extract_fields(bmin, bmax, res, t)
    assert isinstance(u, np.ndarray) and u.shape == (res, res, res) and u.dtype == np.float32
    assert _q(u, g["fields64/u"], 1.0) <= 3 * _q(g["fields/u"], g["fields64/u"], 1.0) + 2e-6
    u2 = r.extract_fields(bmin.cuda(), bmax.cuda(), res, t.cuda(), net_chunk=3 * res * res)      # several launches
    assert np.array_equal(u, u2)
    v, f = r.extract_observation_geometry(t, bmin, bmax, res)
    assert v.ndim == 2 and v.shape[1] == 3 and f.shape[1] == 3 and len(f) > 50
    assert (v >= g["fields/bmin"] - 1e-6).all() and (v <= g["fields/bmax"] + 1e-6).all()
    sdf_at_v = r.sdf_observed(torch.from_numpy(v.astype(np.float32)).cuda(), t.cuda())
    h = float(((g["fields/bmax"] - g["fields/bmin"]) / (res - 1)).max())
    assert float(sdf_at_v.abs().max()) < 0.5 * h                       # vertices sit on the zero level set up to grid resolution


def test_extract_fields_large_grid_matches_point_queries():
    c = load_case("trained_deform")
    r = renderer_for_case(c)
    R = 96
    bmin, bmax, t = torch.tensor([-1.0, -1.0, -1.0]), torch.tensor([1.0, 1.0, 1.0]), torch.tensor([0.5])
    u = r.extract_fields(bmin, bmax, R, t, net_chunk=200000)
    ax = torch.linspace(-1, 1, R)
    idx = torch.randint(0, R, (500, 3), generator=torch.Generator().manual_seed(0))
    pts = torch.stack([ax[idx[:, 0]], ax[idx[:, 1]], ax[idx[:, 2]]], -1).cuda()
    ref = r.sdf_observed(pts, t.cuda()).cpu().numpy()[:, 0]
    got = u[idx[:, 0], idx[:, 1], idx[:, 2]]
    assert np.abs(got - ref).max() < 2e-6
