"""Pin the oracle's offline helpers (renderonpts, renderondepth, extract_fields) against vectors captured from the
reference (tests/golden/offline_*.npz, tools/make_golden_offline.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle_util import T, load_case, oracle_for
from test_oracle_golden import maxdiff, quantile_diff

OFFLINE = ["trained_deform", "trained_nodeform"]


def load_offline(name):
    return load_case(name), load_case("offline_" + name)


@pytest.mark.parametrize("name", OFFLINE)
@pytest.mark.parametrize("dtype,tag", [(torch.float32, ""), (torch.float64, "64")])
def test_renderonpts(name, dtype, tag):
    c, g = load_offline(name)
    R, _ = oracle_for(c, dtype)
    tol = 3e-5 if dtype == torch.float32 else 2e-6       # fp64 goldens are stored as fp32
    x, d, t = T(c["pt/x"], dtype), T(c["pt/d"], dtype), T(c["pt/t"], dtype)
    with torch.no_grad():
        color, normal = R.renderonpts(x, d, t)
        assert quantile_diff(color, g[f"onpts{tag}/color"], 0.99) < tol
        assert quantile_diff(normal, g[f"onpts{tag}/normal"], 0.99) < 10 * tol
        color, normal = R.renderonpts(x.reshape(8, -1, 3), d.reshape(8, -1, 3), torch.tensor([0.37], dtype=dtype))
        assert tuple(color.shape) == g[f"onpts1{tag}/color"].shape == (8, 24, 3)
        assert quantile_diff(color, g[f"onpts1{tag}/color"], 0.99) < tol
        assert quantile_diff(normal, g[f"onpts1{tag}/normal"], 0.99) < 10 * tol
    if dtype == torch.float64:
        assert maxdiff(color, g["onpts164/color"]) < 1e-5


@pytest.mark.parametrize("name", OFFLINE)
def test_renderondepth(name):
    c, g = load_offline(name)
    R, _ = oracle_for(c, torch.float64)
    rays, depth = T(c["rays"], torch.float64), T(g["ondepth/depth_in"], torch.float64)
    with torch.no_grad():
        col, grad, d_out = R.renderondepth(rays, depth)
    valid = (g["ondepth/depth_in"][:, 0] > 0) & np.isfinite(g["ondepth/depth_in"][:, 0])
    assert 0 < valid.sum() < valid.size
    assert maxdiff(col, g["ondepth64/color"]) < 1e-5 and maxdiff(grad, g["ondepth64/gradients"]) < 2e-4
    assert np.all(col.numpy()[~valid] == 0) and np.all(grad.numpy()[~valid] == 0)
    assert maxdiff(d_out, g["ondepth64/d_out"]) < 1e-6 and np.isfinite(d_out.numpy()).all()
    # the reference's own fp32 run agrees with its fp64 run to the same budget (the GPU tests use 3x this)
    assert quantile_diff(g["ondepth/color"], g["ondepth64/color"], 0.99) < 3e-5
    with torch.no_grad():
        col, grad, d_out = R.renderondepth(rays, torch.zeros_like(depth))
    assert float(col.abs().max()) == 0 and maxdiff(d_out, g["ondepth_none64/d_out"]) == 0


@pytest.mark.parametrize("name", OFFLINE)
def test_extract_fields(name):
    c, g = load_offline(name)
    R, _ = oracle_for(c, torch.float64)
    res = int(g["meta/res"])
    with torch.no_grad():
        u = R.extract_fields(T(g["fields/bmin"], torch.float64), T(g["fields/bmax"], torch.float64), res, float(g["fields/t"]))
    assert tuple(u.shape) == (res, res, res) == g["fields64/u"].shape
    assert maxdiff(u, g["fields64/u"]) < 2e-6
    assert maxdiff(g["fields/u"], g["fields64/u"]) < 2e-5       # reference fp32 vs fp64
    assert (g["fields64/u"] < 0).any() and (g["fields64/u"] > 0).any()      # the level set crosses the grid
