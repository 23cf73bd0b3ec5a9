"""Pin the CPU oracle (oracle/endosurf_oracle.py) against vectors captured from the reference
implementation (tests/golden/*.npz, produced by tools/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import endosurf_oracle as O
from oracle_util import CASES, T, load_case, oracle_for


def _np(a):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.asarray(a, np.float64)


def maxdiff(a, b):
    return float(np.max(np.abs(_np(a) - _np(b))))


def quantile_diff(a, b, q):
    return float(np.quantile(np.abs(_np(a) - _np(b)), q))


@pytest.mark.parametrize("name", CASES)
def test_point_networks(name):
    c = load_case(name)
    R, _ = oracle_for(c)
    x, d, t = T(c["pt/x"]), T(c["pt/d"]), T(c["pt/t"])
    with torch.no_grad():
        pe = R.net.point_eval(x, d, t)
        if bool(c["meta/use_deform"]):
            dx, J = R.net.deform(x, t)
            assert maxdiff(dx, c["pt/deform"]) < 2e-6
            # the Jacobian is piecewise constant: allow isolated ReLU-kink flips (hard part 4)
            assert quantile_diff(J, c["pt/J"], 0.995) < 2e-5
        assert maxdiff(R.net.sdf_observed(x, t), c["pt/sdf_observed"]) < 5e-6
    assert maxdiff(pe["sdf"], c["pt/sdf"]) < 5e-6
    assert maxdiff(pe["feat"], c["pt/feat"]) < 2e-5
    assert maxdiff(pe["g_c"], c["pt/g_c"]) < 5e-5
    assert quantile_diff(pe["g_o"], c["pt/g_o"], 0.99) < 5e-5
    assert quantile_diff(pe["rgb"], c["pt/rgb"], 0.99) < 2e-5


@pytest.mark.parametrize("name", CASES)
def test_sampling_trace(name):
    c = load_case(name)
    R, _ = oracle_for(c)
    rays = T(c["rays"])
    u = T(c["u_perturb"]) if "u_perturb" in c else None
    near, far = O.sphere_intersection(rays[:, :3], rays[:, 3:6])
    assert maxdiff(near, c["near"]) < 1e-6 and maxdiff(far, c["far"]) < 1e-6
    with torch.no_grad():
        z, sd, trace = R.sample_z(rays, int(c["meta/iter_step"]), u)
    assert abs(sd - 2.0 / 32) < 1e-12
    for i, zt in enumerate(trace):
        ref = c[f"z_trace/{i}"]
        assert zt.shape == ref.shape
        # inverse-CDF sampling amplifies fp32 noise where the CDF is flat: quantile + loose max
        assert quantile_diff(zt, ref, 0.99) < 2e-4, i
        assert maxdiff(zt, ref) < 5e-3, i


@pytest.mark.parametrize("name", CASES)
def test_render_rays(name):
    c = load_case(name)
    R, _ = oracle_for(c)
    rays = T(c["rays"])
    u = T(c["u_perturb"]) if "u_perturb" in c else None
    with torch.no_grad():
        ret = R.render_rays(rays, int(c["meta/iter_step"]), u)
    # tolerances = ~3x the reference's own fp32-vs-fp64 noise on these cases (DESIGN.md "Tolerances")
    assert maxdiff(ret["color_map"], c["render/color_map"]) < 1e-4
    assert maxdiff(ret["depth_map"], c["render/depth_map"]) < 2e-4
    assert maxdiff(ret["gradient_o_error"], c["render/gradient_o_error"]) < 2e-5 * max(1.0, float(c["render/gradient_o_error"]))
    assert maxdiff(ret["s_val"], c["render/s_val"]) < 1e-7
    assert maxdiff(ret["weight_max"], c["render/weight_max"]) < 3e-4
    for k in ("weights", "cdf"):
        assert quantile_diff(ret[k], c[f"render/{k}"], 0.999) < 6e-4, k
        assert maxdiff(ret[k], c[f"render/{k}"]) < 2e-2, k
    assert quantile_diff(ret["gradients_o"], c["render/gradients_o"], 0.99) < 1e-3
    for k in ("color_map", "depth_map", "gradients_o", "weights", "weight_max", "cdf", "s_val"):
        assert tuple(ret[k].shape) == c[f"render/{k}"].shape


@pytest.mark.parametrize("name", CASES)
def test_aux_losses(name):
    c = load_case(name)
    R, _ = oracle_for(c)
    rays = T(c["rays"])
    with torch.no_grad():
        se, ae, inside = R.errorondepth(rays, T(c["target/depth"]), T(c["target/mask"]))
        d_i = R.ray_marching(rays)
    assert maxdiff(se, c["eod/sdf_error"]) < 1e-5
    assert maxdiff(ae, c["eod/angle_error"]) < 1e-5
    assert maxdiff(inside, c["eod/inside"]) == 0.0
    ref = c["march/d_i"]
    assert np.array_equal(np.isinf(d_i.numpy()), np.isinf(ref))
    fin = np.isfinite(ref)
    assert maxdiff(d_i.numpy()[fin], ref[fin]) < 5e-4
    with torch.no_grad():
        sn, _, valid = R.surface_neighbour_error(rays, T(c["target/mask"]), 0.1, T(c["u_neigh"]))
    assert int(valid.sum()) == int(c["march/n_valid"])
    assert maxdiff(sn, c["sn/value"]) < 3e-4


def _check_grads(params, c, prefix, rtol_norm, atol_frac):
    for name, p in params.items():
        g = p.grad
        if g is None:
            g = torch.zeros_like(p)
        g = g.detach().numpy().astype(np.float64).reshape(-1)
        ref_norm = float(c[f"{prefix}/{name}/norm"])
        norm = float(np.linalg.norm(g))
        assert abs(norm - ref_norm) <= rtol_norm * max(ref_norm, 1e-6) + 1e-7, (name, norm, ref_norm)
        idx, val = c[f"{prefix}/{name}/idx"], c[f"{prefix}/{name}/val"]
        scale = ref_norm / np.sqrt(g.size) + 1e-9
        assert np.max(np.abs(g[idx] - val)) <= atol_frac * scale + 1e-7, (name, np.max(np.abs(g[idx] - val)), scale)


# fp64: the reference run in float64 pins the oracle tightly (no ReLU-kink / rounding noise).
# fp32: the reference's own fp32 parameter gradients deviate from its fp64 ones by up to ~7 % on the
# trained cases (isolated kink flips of the deformation Jacobian), so fp32-vs-fp32 can only be loose.
GRAD_TOL = {torch.float64: dict(rtol_norm=2e-4, atol_frac=2e-2), torch.float32: dict(rtol_norm=0.12, atol_frac=6.0)}


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", CASES)
def test_training_loss_and_param_grads(name, dtype):
    c = load_case(name)
    tag = "64" if dtype == torch.float64 else ""
    R, params = oracle_for(c, dtype, requires_grad=True)
    batch = dict(rays=T(c["rays"], dtype), color=T(c["target/color"], dtype), depth=T(c["target/depth"], dtype),
                 mask=T(c["target/mask"], dtype), color_mask=T(c["target/color_mask"], dtype))
    u = T(c["u_perturb"], dtype) if "u_perturb" in c else None
    total, terms, _ = O.train_loss(R, batch, int(c["meta/iter_step"]), u, T(c["u_neigh"], dtype))
    tol = 1e-6 if dtype == torch.float64 else 2e-4
    for k, v in terms.items():
        assert maxdiff(v, c[f"loss{tag}/{k}"]) < tol * max(1.0, abs(float(c[f"loss{tag}/{k}"]))), k
    assert maxdiff(total, c[f"loss{tag}/total"]) < tol * max(1.0, abs(float(c[f"loss{tag}/total"])))
    total.backward()
    _check_grads(params, c, f"grad{tag}", **GRAD_TOL[dtype])


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", CASES)
def test_render_scalar_param_grads(name, dtype):
    c = load_case(name)
    tag = "64" if dtype == torch.float64 else ""
    R, params = oracle_for(c, dtype, requires_grad=True)
    rays = T(c["rays"], dtype)
    u = T(c["u_perturb"], dtype) if "u_perturb" in c else None
    ret = R.render_rays(rays, int(c["meta/iter_step"]), u)
    cw, dw, gw, ww = (T(c[f"scal/{k}"], dtype) for k in ("cw", "dw", "gw", "ww"))
    scal = ((ret["color_map"] * cw).sum() + (ret["depth_map"] * dw).sum() + (ret["gradients_o"] * gw).sum()
            + (ret["weights"] * ww).sum() + 0.5 * ret["gradient_o_error"] + (ret["cdf"] * ww).sum() * 0.1
            + ret["s_val"].sum() * 0.01)
    tol = 1e-6 if dtype == torch.float64 else 1e-3
    assert maxdiff(scal, c[f"scal{tag}/value"]) < tol * max(1.0, abs(float(c[f"scal{tag}/value"])))
    scal.backward()
    _check_grads(params, c, f"scalgrad{tag}", **GRAD_TOL[dtype])


@pytest.mark.parametrize("name", CASES)
def test_fp64_forward_tight(name):
    """Oracle in fp64 vs the reference run in fp64: tight pin of every forward output."""
    c = load_case(name)
    dt = torch.float64
    R, _ = oracle_for(c, dt)
    rays = T(c["rays"], dt)
    u = T(c["u_perturb"], dt) if "u_perturb" in c else None
    with torch.no_grad():
        ret = R.render_rays(rays, int(c["meta/iter_step"]), u)
        pe = R.net.point_eval(T(c["pt/x"], dt), T(c["pt/d"], dt), T(c["pt/t"], dt))
        d_i = R.ray_marching(rays)
    # fixtures hold the fp64 results rounded to fp32 (6e-8 relative)
    for k, tol in dict(color_map=1e-6, depth_map=1e-6, weights=1e-6, cdf=1e-6, gradients_o=4e-6, weight_max=1e-6).items():
        assert maxdiff(ret[k], c[f"render64/{k}"]) < tol, (k, maxdiff(ret[k], c[f"render64/{k}"]))
    for i, z in enumerate(ret["z_trace"]):
        assert maxdiff(z, c[f"z_trace64/{i}"]) < 2e-6, i
    for k, tol in dict(sdf=2e-7, feat=1e-6, g_c=1e-6, g_o=1e-6, rgb=2e-7, J=1e-6).items():
        assert maxdiff(pe[k], c[f"pt64/{k}"]) < tol, (k, maxdiff(pe[k], c[f"pt64/{k}"]))
    ref = c["march64/d_i"]
    assert np.array_equal(np.isinf(d_i.numpy()), np.isinf(ref))
    fin = np.isfinite(ref)
    assert maxdiff(d_i.numpy()[fin], ref[fin]) < 1e-6


@pytest.mark.parametrize("name", ["trained_deform", "init_deform"])
def test_color_network_on_explicit_inputs(name):
    """OracleNet.color vs ColorNetwork.forward(x, n, d, geo_feat) of the reference (tests/golden/color_direct.npz, tools/make_golden_color.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "color_direct.npz"))
    seed, trained, use_deform = (int(v) for v in g[f"{name}/meta"])
    import weightgen
    from oracle import endosurf_oracle as O
    state = weightgen.make_state(seed, "trained" if trained else "init", bool(use_deform))
    for dtype, key, tol in ((torch.float64, "rgb64", 1e-9), (torch.float32, "rgb", 2e-5)):
        net = O.OracleNet({k: torch.tensor(v, dtype=dtype) for k, v in state.items()}, bool(use_deform))
        rgb = net.color(*(torch.tensor(g[f"{name}/{k}"], dtype=dtype) for k in ("x", "n", "d", "feat")))
        assert float((rgb.double() - torch.tensor(g[f"{name}/{key}"], dtype=torch.float64)).abs().max()) < tol


def test_oracle_fp32_gradient_error_matches_the_references():
    """The goldens' whole-tensor ``scalgraderr/*/rel`` (the reference's own fp32-vs-fp64 parameter-gradient error, round 4) is a property of
    the ALGORITHM in fp32, not of one implementation: the oracle run in both precisions shows the same figures (init_deform: 0.2-1.2 % on
    the colour network, where one render-level scalar moves sample positions by fp32 noise).  This pins the field the GPU tests budget
    the HIP gradients against (tests/test_gpu_backward.py:_check_rows)."""
    c = load_case("init_deform")
    it = int(c["meta/iter_step"])
    g = {}
    for dt in (torch.float64, torch.float32):
        R, params = oracle_for(c, dt, requires_grad=True)
        ret = R.render_rays(T(c["rays"], dt), it, None)
        cw, dw, gw, ww = (torch.tensor(c[f"scal/{k}"], dtype=dt) for k in ("cw", "dw", "gw", "ww"))
        scal = ((ret["color_map"] * cw).sum() + (ret["depth_map"] * dw).sum() + (ret["gradients_o"] * gw).sum() + (ret["weights"] * ww).sum()
                + 0.5 * ret["gradient_o_error"] + (ret["cdf"] * ww).sum() * 0.1 + ret["s_val"].sum() * 0.01)
        scal.backward()
        g[dt] = {k: p.grad.detach().double().reshape(-1) for k, p in params.items()}
    n = 0
    for k, g64 in g[torch.float64].items():
        ref = float(c[f"scalgraderr/{k}/rel"])
        if ref < 2e-3:
            continue
        own = float((g[torch.float32][k] - g64).norm() / (g64.norm() + 1e-300))
        assert 0.5 * ref < own < 2.0 * ref, (k, own, ref)
        n += 1
    assert n >= 15, n
