"""GPU parity of the drop-in's PUBLIC reference methods that round 1 lacked: EndoSurfRenderer.up_sample /
cat_z_vals / secant (endosurf.py:221-287, 422-449) and the EndoSurfNet query surface (endosurf.py:570-689), against vectors the
reference itself produced (tests/golden/*.npz, tools/make_golden.py)."""
import numpy as np
import pytest
import torch

from gpu_util import renderer_for, renderer_for_case
from oracle_util import CASES, load_case

pytestmark = pytest.mark.gpu


def qdiff(a, b, q):
    return float(np.quantile(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)), q))


@pytest.mark.parametrize("name", CASES)
def test_up_sample_and_cat_z_vals_follow_reference_trace(name):
    """Drive up_sample + cat_z_vals exactly as tools/make_golden.py:z_trace drives the reference's and compare every
    intermediate z / sdf array (budget: 3x the reference's own fp32-vs-fp64 difference)."""
    c = load_case(name)
    r = renderer_for_case(c)
    rays = torch.from_numpy(c["rays"]).cuda()
    o, d, time = rays[:, :3], rays[:, 3:6], rays[:, 8]
    z = torch.from_numpy(c["z_trace/0"]).cuda()
    with torch.no_grad():
        sdf = r.model.get_sdf_from_observed_space(
            (o[:, None, :] + (d / (d[:, 2:] + 1e-6))[:, None, :] * z[:, :, None]).reshape(-1, 3),
            time[:, None, None].expand(z.shape[0], z.shape[1], 1).reshape(-1, 1)).reshape(z.shape)
    assert np.max(np.abs(sdf.cpu().numpy() - c["sdf_trace64/0"])) < 3 * np.max(np.abs(c["sdf_trace/0"] - c["sdf_trace64/0"])) + 1e-5
    for i in range(r.up_sample_steps):
        new_z = r.up_sample(o, d, z, sdf, r.n_importance // r.up_sample_steps, 64 * 2 ** i)
        assert tuple(new_z.shape) == (z.shape[0], r.n_importance // r.up_sample_steps)
        last = i + 1 == r.up_sample_steps
        z, sdf = r.cat_z_vals(o, d, time, z, new_z, sdf, last=last)
        zt = z.cpu().numpy()
        ref64, ref32 = c[f"z_trace64/{i + 1}"], c[f"z_trace/{i + 1}"]
        assert zt.shape == ref64.shape and np.all(np.diff(zt, axis=1) >= 0)
        assert qdiff(zt, ref64, 0.99) < 3 * qdiff(ref32, ref64, 0.99) + 2e-6, i
        assert np.max(np.abs(zt - ref64)) < 3 * np.max(np.abs(ref32 - ref64)) + 1e-4, i
        if not last:
            s64, s32 = c[f"sdf_trace64/{i + 1}"], c[f"sdf_trace/{i + 1}"]
            st = sdf.cpu().numpy()
            assert st.shape == s64.shape
            assert qdiff(st, s64, 0.99) < 3 * qdiff(s32, s64, 0.99) + 1e-5, i
    # the public methods give what the fused sampling stage gives
    u = torch.from_numpy(c["u_perturb"]).cuda() if "u_perturb" in c else None
    zs = r.sample_z(rays, int(c["meta/iter_step"]), perturb_overwrite=u is not None, u_perturb=u)
    assert np.max(np.abs(zs.cpu().numpy() - z.cpu().numpy())) < 3 * np.max(np.abs(c["z_trace/4"] - c["z_trace64/4"])) + 1e-4


@pytest.mark.parametrize("name", CASES)
def test_public_secant_matches_ray_marching(name):
    """ray_marching == (first sign change) + secant: feed the public secant the bracket the reference's ray_marching would pass
    it (endosurf.py:399-413) and compare with march64/d_i."""
    c = load_case(name)
    r = renderer_for_case(c)
    rays = torch.from_numpy(c["rays"]).cuda()
    N, n_steps = rays.shape[0], 128
    dprop = r.engine.empty(N, n_steps)
    near, far = r.engine.ray_setup(rays, None, n_steps, 0.0, 1, dprop, want_bounds=True)
    o, d, time = rays[:, :3], rays[:, 3:6], rays[:, 8]
    pts = (o[:, None, :] + (d / (d[:, 2:] + 1e-6))[:, None, :] * dprop[:, :, None]).reshape(-1, 3)
    with torch.no_grad():
        val = -r.model.get_sdf_from_observed_space(pts, time[:, None].expand(N, n_steps).reshape(-1)).reshape(N, n_steps)
    # first sign change, as the reference finds it
    sign = torch.cat([torch.sign(val[:, :-1] * val[:, 1:]), torch.ones(N, 1, device="cuda")], -1)
    cost = sign * torch.arange(n_steps, 0, -1, device="cuda").float()
    values, idx = torch.min(cost, -1)
    ar = torch.arange(N, device="cuda")
    mask = (values < 0) & (val[ar, idx] < 0) & (val[:, 0] < 0)
    idx2 = torch.clamp(idx + 1, max=n_steps - 1)
    d_low, f_low, d_high, f_high = dprop[ar, idx][mask], val[ar, idx][mask], dprop[ar, idx2][mask], val[ar, idx2][mask]
    d_pred = r.secant(f_low, f_high, d_low, d_high, 8, rays[mask], 0.0, 64000)
    ref64, ref32 = c["march64/d_i"][:, 0], c["march/d_i"][:, 0]
    fin = np.isfinite(ref64) & (ref64 != 0)
    assert np.array_equal(mask.cpu().numpy(), fin)
    budget = 3 * np.max(np.abs(ref32[fin] - ref64[fin])) + 2e-5
    assert np.max(np.abs(d_pred.cpu().numpy() - ref64[fin])) < budget
    # and the fused device-side ray_marching agrees with it (both sit within the budget of the reference's fp64 result; the secant
    # iteration amplifies the rounding differences of the two SDF query kernels they use)
    d_i = r.ray_marching(rays).cpu().numpy()[:, 0]
    assert np.max(np.abs(d_i[fin] - d_pred.cpu().numpy())) < 2 * budget


@pytest.mark.parametrize("name", CASES)
def test_model_query_surface(name):
    """renderer.model.{get_sdf_from_observed_space, get_sdf_grad_from_observed_space, get_sdf_grad_from_canonical_space,
    get_deform_grad_from_observed_space, forward} against the reference's pt64/* vectors."""
    c = load_case(name)
    r = renderer_for_case(c)
    m = r.model
    x, d, t = (torch.from_numpy(c[f"pt/{k}"]).cuda() for k in ("x", "d", "t"))
    g = lambda a: a.detach().cpu().numpy()
    with torch.no_grad():
        assert np.max(np.abs(g(m.get_sdf_from_observed_space(x, t)) - c["pt64/sdf_observed"])) < 1e-5
        assert np.max(np.abs(g(m.get_sdf_grad_from_observed_space(x, t)) - c["pt64/g_o"])) < 1e-4
        J = g(m.get_deform_grad_from_observed_space(x, t))
        assert J.shape == c["pt64/J"].shape == (x.shape[0], 3, 3)
        assert np.max(np.abs(J - c["pt64/J"])) < 1e-4
        x_c = x + torch.from_numpy(c["pt64/deform"]).cuda() if r.use_deform else x
        assert np.max(np.abs(g(m.get_sdf_grad_from_canonical_space(x_c)) - c["pt64/g_c"])) < 1e-4
        out = g(m.forward(torch.cat([x, d, t], -1)))
        assert out.shape == (x.shape[0], 4)
        assert np.max(np.abs(out[:, :1] - c["pt64/sdf_observed"])) < 1e-5
        assert np.max(np.abs(out[:, 1:] - c["pt64/rgb"])) < 5e-5
    # grad mode on: the same values, differentiable w.r.t. the parameters (hand-written backward)
    for p in r.parameters():
        p.grad = None
    sdf = m.get_sdf_from_observed_space(x, t)
    assert sdf.requires_grad and np.max(np.abs(g(sdf) - c["pt64/sdf_observed"])) < 1e-5
    (sdf.sum() + m.forward(torch.cat([x, d, t], -1))[:, 1:].sum()).backward()
    gn = dict(r.named_parameters())["model.sdf_network.net.2.weight_v"].grad
    assert gn is not None and float(gn.norm()) > 0


@pytest.mark.parametrize("mode,use_deform", [("trained", True), ("trained", False)])
def test_sdf_query_is_differentiable_wrt_points_first_order(mode, use_deform):
    """The reference's own pattern around get_sdf_from_observed_space (endosurf.py:585-600): ``x.requires_grad_(True); sdf = model(x);
    g = autograd.grad(sdf, x, create_graph=True)`` and a loss on g (an eikonal term on arbitrary points) back-propagated to the
    parameters.  HIP: d sdf / d x = g_o of the fused kernels, whose own backward carries d g_o / d theta.  Checked against the fp64
    oracle (autograd through the closed-form restatement): g, the loss, and the parameter gradients of sdf AND of the loss on g."""
    import weightgen
    from oracle import endosurf_oracle as O
    seed, M = 11, 192
    r = renderer_for(seed, mode, use_deform)
    state = weightgen.make_state(seed, mode, use_deform)
    params = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in state.items()}
    net = O.OracleNet(params, use_deform)
    rng = np.random.default_rng(5)
    x_np = rng.uniform(-0.8, 0.8, size=(M, 3)).astype(np.float32)
    t_np = rng.uniform(0.0, 1.0, size=(M, 1)).astype(np.float32)

    def run(sdf_fn, x, t):
        sdf = sdf_fn(x, t)
        (g,) = torch.autograd.grad(sdf, x, grad_outputs=torch.ones_like(sdf), create_graph=True)
        loss = ((g.norm(dim=-1) - 1.0) ** 2).mean() + 0.3 * sdf.mean()
        loss.backward()
        return sdf.detach(), g.detach(), float(loss.detach())

    x64 = torch.from_numpy(x_np).double().requires_grad_(True)
    o_sdf, o_g, o_loss = run(net.sdf_observed, x64, torch.from_numpy(t_np).double())
    for p in r.parameters():
        p.grad = None
    x32 = torch.from_numpy(x_np).cuda().requires_grad_(True)
    h_sdf, h_g, h_loss = run(r.model.get_sdf_from_observed_space, x32, torch.from_numpy(t_np).cuda())
    assert float((h_sdf.double().cpu() - o_sdf).abs().max()) < 1e-5
    assert float((h_g.double().cpu() - o_g).abs().max()) < 1e-4
    assert abs(h_loss - o_loss) < 1e-5 * max(1.0, abs(o_loss))
    # the gradient w.r.t. the points themselves: the sdf term's share (0.3 / M g_o) AND, since round 5, the eikonal term's -- the
    # Hessian-vector product of the query that the reference gets from create_graph=True (Engine.point_input_adjoint)
    xg_h, xg_o = x32.grad.double().cpu(), x64.grad
    assert float((xg_h - xg_o).norm() / xg_o.norm()) < 1e-4, float((xg_h - xg_o).norm() / xg_o.norm())
    assert float((xg_o - 0.3 / M * o_g).norm() / xg_o.norm()) > 0.1          # (the second-order share is not negligible in this check)
    named = dict(r.named_parameters())
    keys = ["sdf_network.net.0.weight_v", "sdf_network.net.4.weight_g", "sdf_network.net.7.bias"] + (
        ["deform_network.net.1.weight_v", "deform_network.net.6.bias"] if use_deform else [])
    for k in keys:
        gh, go = named["model." + k].grad.double().cpu(), params[k].grad
        rel = float((gh - go).norm() / (go.norm() + 1e-30))
        assert rel < 2e-3, (k, rel)


@pytest.mark.parametrize("mode,use_deform,canonical", [("trained", True, False), ("init", True, False), ("trained", False, False),
                                                       ("trained", True, True)])
def test_sdf_gradient_query_is_differentiable_wrt_points(mode, use_deform, canonical):
    """get_sdf_grad_from_observed_space / get_sdf_grad_from_canonical_space return their gradient with create_graph=True in the reference
    (endosurf.py:598, :616): ``autograd.grad(<g, w>, x)`` is the Hessian of the query times w.  HIP: J^T xcbar - w * curv(g_c)
    (es_point_vjp on the x_c adjoint of es_point_backward; xcbar itself without a deformation network).  Against autograd through the
    fp64 oracle, 1e-4 relative, together with the parameter gradients of the same backward; and no RuntimeWarning about detached points."""
    import warnings
    import weightgen
    from oracle import endosurf_oracle as O
    seed, M = 13, 200
    r = renderer_for(seed, mode, use_deform)
    state = weightgen.make_state(seed, mode, use_deform)
    params = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in state.items()}
    net = O.OracleNet(params, use_deform)
    rng = np.random.default_rng(6)
    x_np = rng.uniform(-0.8, 0.8, size=(M, 3)).astype(np.float32)
    t_np = rng.uniform(0.0, 1.0, size=(M, 1)).astype(np.float32)
    w_np = rng.normal(size=(M, 3)).astype(np.float32)

    x64 = torch.from_numpy(x_np).double().requires_grad_(True)
    if canonical:
        sdf64 = net.sdf_net(x64, with_grad=False)[0]
    else:
        sdf64 = net.sdf_observed(x64, torch.from_numpy(t_np).double())
    (g64,) = torch.autograd.grad(sdf64.sum(), x64, create_graph=True)
    (g64 * torch.from_numpy(w_np).double()).sum().backward()

    for p in r.parameters():
        p.grad = None
    x32 = torch.from_numpy(x_np).cuda().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        g32 = (r.model.get_sdf_grad_from_canonical_space(x32) if canonical
               else r.model.get_sdf_grad_from_observed_space(x32, torch.from_numpy(t_np).cuda()))
    assert g32.requires_grad and float((g32.detach().double().cpu() - g64.detach()).abs().max()) < 1e-4
    (g32 * torch.from_numpy(w_np).cuda()).sum().backward()
    xh, xo = x32.grad.double().cpu(), x64.grad
    rel = float((xh - xo).norm() / xo.norm())
    assert rel < 1e-4, rel
    named = dict(r.named_parameters())
    keys = ["sdf_network.net.0.weight_v", "sdf_network.net.4.weight_g", "sdf_network.net.7.bias"] + (
        ["deform_network.net.1.weight_v", "deform_network.net.6.bias"] if use_deform and not canonical else [])
    for k in keys:
        gh, go = named["model." + k].grad.double().cpu(), params[k].grad
        assert float((gh - go).norm() / (go.norm() + 1e-30)) < 2e-3, k


def test_parameter_rebinding_is_detected():
    """Anything that gives a parameter its own storage (``p.data = ...`` loaders) is folded back into the flat buffer the kernels
    read, and ``.to()`` moves the flat buffer itself (stale-weights hazard)."""
    c = load_case("trained_deform")
    r = renderer_for_case(c)
    x, t = torch.from_numpy(c["pt/x"]).cuda(), torch.from_numpy(c["pt/t"]).cuda()
    with torch.no_grad():
        s0 = r.model.get_sdf_from_observed_space(x, t).clone()
        p = r.model.sdf_network.net[8].bias
        p.data = p.data.clone() + 0.25          # re-bound storage: no longer a view of model._flat
        s1 = r.model.get_sdf_from_observed_space(x, t)
        assert torch.allclose(s1, s0 + 0.25, atol=1e-5)
        assert p.data_ptr() == r.model._flat.data_ptr() + 4 * r.model._layout["sdf_network.net.8.bias"][0]
        r.to("cuda")
        assert torch.allclose(r.model.get_sdf_from_observed_space(x, t), s1, atol=0)
        # a Parameter OBJECT assigned anew (not only its storage): the cached parameter walk is rebuilt and the value folded back
        lin = r.model.sdf_network.net[8]
        lin.bias = torch.nn.Parameter(lin.bias.detach().clone() - 0.25)
        s2 = r.model.get_sdf_from_observed_space(x, t)
        assert torch.allclose(s2, s0, atol=1e-5)
        assert lin.bias.data_ptr() == r.model._flat.data_ptr() + 4 * r.model._layout["sdf_network.net.8.bias"][0]
        with pytest.raises(TypeError):
            r.double()


def test_surface_neighbour_error_documented_divergences():
    """Two conscious divergences from endosurf.py:319-342, pinned here so that they cannot drift silently:
    (1) with no valid ray the reference returns the python float 0.; the drop-in returns a 0-d tensor equal to 0 (fixed shape,
        no host synchronisation) that still supports ``0.1 * sn`` and ``.backward()`` of a sum containing it;
    (2) the neighbour offsets: the reference draws rand_like(p_surf[valid]) (n_valid x 3 numbers, the k-th valid ray uses the
        k-th row); the drop-in draws one row per RAY ([N,3], ray i uses row i).  With explicit ``u_neigh`` [N,3] the value equals the
        reference's when it is handed u_neigh[valid] (that is how tools/make_golden.py feeds it: sn64/value)."""
    c = load_case("trained_deform")
    r = renderer_for_case(c)
    rays = torch.from_numpy(c["rays"]).cuda()
    mask = torch.from_numpy(c["target/mask"]).cuda()
    sn0 = r.surface_neighbour_error(rays, torch.zeros_like(mask), neighbour_rad=0.1)
    assert torch.is_tensor(sn0) and sn0.dim() == 0 and float(sn0) == 0.0
    (0.1 * sn0 + sum(p.sum() * 0 for p in r.parameters())).backward()
    un = torch.from_numpy(c["u_neigh"]).cuda()
    sn = r.surface_neighbour_error(rays, mask, neighbour_rad=0.1, u_neigh=un)
    v64, v32 = float(c["sn64/value"]), float(c["sn/value"])
    assert abs(float(sn) - v64) < 3 * abs(v32 - v64) + 2e-5 * max(1.0, abs(v64))
    # rows of rays without a valid hit are never consumed: scrambling them changes nothing
    with torch.no_grad():
        d_i = r.ray_marching(rays)
    valid = (torch.isfinite(d_i) & (d_i != 0) & (mask == 1))[:, 0]
    un2 = torch.where(valid[:, None], un, torch.rand_like(un))
    assert float(r.surface_neighbour_error(rays, mask, neighbour_rad=0.1, u_neigh=un2)) == float(sn)


@pytest.mark.parametrize("name", CASES)
def test_sub_network_forwards(name):
    """renderer.model.{deform_network(x, t), sdf_network(x_c), sdf_network.sdf(x_c), deviation_network(x)} (endosurf.py:724-852)
    against the reference's pt64/* vectors."""
    c = load_case(name)
    r = renderer_for_case(c)
    m = r.model
    x, t = torch.from_numpy(c["pt/x"]).cuda(), torch.from_numpy(c["pt/t"]).cuda()
    if r.use_deform:
        dx = m.deform_network(x, t)
        assert np.max(np.abs(dx.cpu().numpy() - c["pt64/deform"])) < 1e-5
        x_c = x + torch.from_numpy(c["pt64/deform"]).cuda()
    else:
        x_c = x
    h = m.sdf_network(x_c).cpu().numpy()
    assert h.shape == (x.shape[0], 257)
    assert np.max(np.abs(h[:, :1] - c["pt64/sdf"])) < 1e-5 and np.max(np.abs(h[:, 1:] - c["pt64/feat"])) < 5e-5
    assert np.max(np.abs(m.sdf_network.sdf(x_c).cpu().numpy() - c["pt64/sdf"])) < 1e-5
    inv_s = m.deviation_network(x)
    assert tuple(inv_s.shape) == (x.shape[0], 1) and inv_s.requires_grad
    assert abs(float(inv_s[0, 0]) - float(np.exp(10.0 * float(m.deviation_network.variance)))) < 1e-3
    # the colour network on explicit inputs: test_color_network_forward_on_explicit_inputs below


@pytest.mark.parametrize("name", ["trained_deform", "init_deform"])
def test_color_network_forward_on_explicit_inputs(name):
    """renderer.model.color_network(x, n, d, geo_feat) = the reference's ColorNetwork.forward on explicit inputs (endosurf.py:828-842):
    d is taken as given (NOT normalised), M is not a multiple of the tile.  Budget: 3x the reference's own fp32-vs-fp64 error."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "color_direct.npz"))
    seed, trained, use_deform = (int(v) for v in g[f"{name}/meta"])
    r = renderer_for(seed, "trained" if trained else "init", bool(use_deform))
    x, n, d, feat = (torch.from_numpy(g[f"{name}/{k}"]).cuda() for k in ("x", "n", "d", "feat"))
    rgb = r.model.color_network(x, n, d, feat)
    assert tuple(rgb.shape) == (x.shape[0], 3)
    ref64 = g[f"{name}/rgb64"]
    budget = 3 * float(np.abs(g[f"{name}/rgb"].astype(np.float64) - ref64).max()) + 2e-6
    assert float(np.abs(rgb.double().cpu().numpy() - ref64).max()) <= budget
    # a scaled direction gives a different colour (the network does not normalise it) and the leading shape is free
    rgb2 = r.model.color_network(x.reshape(4, 50, 3), n.reshape(4, 50, 3), 2.0 * d.reshape(4, 50, 3), feat.reshape(4, 50, 256))
    assert tuple(rgb2.shape) == (200, 3) and float((rgb2 - rgb).abs().max()) > 1e-4
    assert r.model.color_network(x[:0], n[:0], d[:0], feat[:0]).shape == (0, 3)


def test_point_adjoint_keeps_the_callers_shape_and_dtype():
    """The adjoint of the query points comes back in the caller's tensor: any leading shape, float64 points, a scalar time."""
    r = renderer_for(13, "trained", True)
    x = (torch.rand(4, 5, 3, device="cuda", dtype=torch.float64) - 0.5).requires_grad_(True)
    g = r.model.get_sdf_grad_from_observed_space(x, torch.tensor(0.3, device="cuda"))
    assert tuple(g.shape) == (20, 3) and g.requires_grad
    (g ** 2).sum().backward()
    assert x.grad is not None and tuple(x.grad.shape) == (4, 5, 3) and x.grad.dtype == torch.float64
    assert bool(torch.isfinite(x.grad).all()) and float(x.grad.abs().max()) > 0
    # twice through the same node (retain_graph): the consumed workspace is re-evaluated, the result is the same
    x2 = x.detach().clone().requires_grad_(True)
    g2 = r.model.get_sdf_grad_from_observed_space(x2, torch.tensor(0.3, device="cuda"))
    l2 = (g2 ** 2).sum()
    (a,) = torch.autograd.grad(l2, x2, retain_graph=True)
    (b,) = torch.autograd.grad(l2, x2)
    assert torch.allclose(a, x.grad, rtol=1e-5, atol=1e-7) and torch.allclose(a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("mode,use_deform", [("trained", True), ("init", True), ("trained", False)])
def test_model_forward_is_differentiable_wrt_its_inputs(mode, use_deform):
    """EndoSurfNet.forward(cat[x, d, t]) -> [sdf, rgb] (endosurf.py:660-689) is an ordinary autograd function of its inputs in the
    reference.  HIP: xbar = J^T xcbar - d * curv(vbar), dbar = J^T vbar, tbar = <xcbar, d x_c / d t> from the backward's adjoints of x_c and
    v = J d and two more reverse sweeps of the deformation network (renderer._NetForwardFn).  Against autograd through the fp64 oracle:
    1e-4 relative on the input gradient, together with the parameter gradients of the same backward."""
    import weightgen
    from oracle import endosurf_oracle as O
    seed, M = 17, 200
    r = renderer_for(seed, mode, use_deform)
    state = weightgen.make_state(seed, mode, use_deform)
    params = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in state.items()}
    net = O.OracleNet(params, use_deform)
    rng = np.random.default_rng(8)
    x = rng.uniform(-0.8, 0.8, size=(M, 3))
    d = rng.normal(size=(M, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d *= rng.uniform(0.5, 2.0, size=(M, 1))            # (forward normalises J d itself: its gradient w.r.t. d is tangential)
    t = rng.uniform(0.0, 1.0, size=(M, 1))
    inp_np = np.concatenate([x, d, t], -1).astype(np.float32)
    w_np = rng.normal(size=(M, 4)).astype(np.float32)

    i64 = torch.from_numpy(inp_np).double().requires_grad_(True)
    o = net.point_eval(i64[:, :3], i64[:, 3:6], i64[:, 6], with_color=True)
    out64 = torch.cat([o["sdf"], o["rgb"]], -1)
    (out64 * torch.from_numpy(w_np).double()).sum().backward()

    for p in r.parameters():
        p.grad = None
    i32 = torch.from_numpy(inp_np).cuda().requires_grad_(True)
    out32 = r.model(i32)
    assert out32.shape == (M, 4) and out32.requires_grad
    assert float((out32.detach().double().cpu() - out64.detach()).abs().max()) < 5e-5
    (out32 * torch.from_numpy(w_np).cuda()).sum().backward()
    gh, go = i32.grad.double().cpu(), i64.grad
    for name, sl in (("x", slice(0, 3)), ("d", slice(3, 6)), ("t", slice(6, 7))):
        ref = go[:, sl]
        if float(ref.norm()) == 0.0:                   # (no deformation network: the output does not depend on the time)
            assert float(gh[:, sl].abs().max()) == 0.0, name
            continue
        rel = float((gh[:, sl] - ref).norm() / ref.norm())
        assert rel < 1e-4, (name, rel)
    named = dict(r.named_parameters())
    keys = ["sdf_network.net.0.weight_v", "sdf_network.net.7.bias", "color_network.net.0.weight_v", "color_network.net.8.bias"] + (
        ["deform_network.net.1.weight_v", "deform_network.net.6.bias"] if use_deform else [])
    for k in keys:
        a, b = named["model." + k].grad.double().cpu(), params[k].grad
        assert float((a - b).norm() / (b.norm() + 1e-30)) < 2e-3, k
    # a second backward through the same node (retain_graph) re-evaluates the consumed workspace
    i32b = torch.from_numpy(inp_np).cuda().requires_grad_(True)
    s = (r.model(i32b) * torch.from_numpy(w_np).cuda()).sum()
    (g1,) = torch.autograd.grad(s, i32b, retain_graph=True)
    (g2,) = torch.autograd.grad(s, i32b)
    assert float((g1 - g2).abs().max()) <= 1e-6 * float(g1.abs().max()) and float((g1 - i32.grad).abs().max()) <= 1e-5 * float(g1.abs().max())


def test_single_layer_call_matches_the_weight_normed_linear():
    """``model.sdf_network.net[l](x)``: one weight-normed nn.Linear of the reference (utils.py:57-58) -- plain torch on the parameter
    views, against the oracle's effective weight; gradients reach the parameters."""
    from oracle import endosurf_oracle as O
    c = load_case("trained_deform")
    r = renderer_for_case(c)
    lin = r.model.sdf_network.net[2]
    x = torch.randn(17, lin.weight_v.shape[1], device="cuda")
    y = lin(x)
    W = O.effective_weight(lin.weight_g.detach().double().cpu(), lin.weight_v.detach().double().cpu())
    ref = x.double().cpu() @ W.t() + lin.bias.detach().double().cpu()
    assert float((y.detach().double().cpu() - ref).abs().max()) < 1e-5
    y.sum().backward()
    assert lin.weight_v.grad is not None and lin.weight_g.grad is not None and float(lin.bias.grad.sum()) == 17 * lin.bias.numel()
