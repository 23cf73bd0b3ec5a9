"""The reference trainer's OWN call sequence through the drop-in (trainer_endosurf.py:130, :140, :155: ``renderer(rays)`` ->
``errorondepth`` -> ``surface_neighbour_error`` as three calls, one ``loss.backward()``): from the second step on the two later calls'
points are evaluated into the tail of the live render's workspace and back-propagated by the render's ONE backward chain
(renderer._Tail).  Same values, same gradients as the stand-alone evaluations -- whatever subset of the three results the loss uses,
when the tail overflows, and when nobody claims it."""
import pytest
import torch

from gpu_util import renderer_for_case
from oracle_util import load_case

pytestmark = pytest.mark.gpu


def _batch(c, dev="cuda"):
    b = dict(rays=torch.from_numpy(c["rays"]).to(dev), color=torch.from_numpy(c["target/color"]).to(dev),
             depth=torch.from_numpy(c["target/depth"]).to(dev), mask=torch.from_numpy(c["target/mask"]).to(dev),
             color_mask=torch.from_numpy(c["target/color_mask"]).to(dev))
    u = torch.from_numpy(c["u_perturb"]).to(dev) if "u_perturb" in c else None
    return b, u, torch.from_numpy(c["u_neigh"]).to(dev)


def _grads(r):
    return {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in r.named_parameters()}


def _step(r, b, u, un, it, use, demand):
    """One pass of the reference call sequence; ``use``: which of the three calls' results enter the loss; ``demand``: tail rows the
    render reserves (0: every evaluation stand-alone)."""
    for p in r.parameters():
        p.grad = None
    r._aux_demand = demand
    r.perturb = u is not None
    ret = r(b["rays"], iter_step=it, u_perturb=u)
    tail = r._live_tail[0] if r._live_tail is not None else None
    sdf_loss, angle_loss, valid = r.errorondepth(b["rays"], d_gt=b["depth"], mask=b["mask"], iter_step=it)
    sn = r.surface_neighbour_error(rays=b["rays"], mask=b["mask"], iter_step=it, neighbour_rad=0.1, u_neigh=un)
    loss = 0.0
    if "render" in use:
        loss = loss + ((ret["color_map"] - b["color"]) * b["color_mask"]).abs().sum() / (b["color_mask"].sum() + 1e-10) + 0.1 * ret["gradient_o_error"]
        loss = loss + ((ret["depth_map"] - b["depth"]) * valid * b["mask"]).abs().sum() / ((valid * b["mask"]).sum() + 1e-10)
    if "eod" in use:
        loss = loss + sdf_loss + 0.1 * angle_loss
    if "sn" in use:
        loss = loss + 0.1 * sn
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), (float(sdf_loss), float(angle_loss), float(sn)), _grads(r), tail


def _close(ga, gb, tol=3e-4):
    for k in ga:
        n = float(ga[k].norm())
        assert float((ga[k] - gb[k]).norm()) <= tol * n + 1e-7, (k, float((ga[k] - gb[k]).norm()), n)


@pytest.mark.parametrize("name", ["trained_deform", "trained_nodeform"])
@pytest.mark.parametrize("use", [("render", "eod", "sn"), ("eod", "sn"), ("sn",), ("render",)])
def test_tail_equals_standalone(name, use):
    c = load_case(name)
    b, u, un = _batch(c)
    it = int(c["meta/iter_step"])
    r = renderer_for_case(c)
    r.engine.deterministic = True
    N = b["rays"].shape[0]
    l0, t0, g0, tail0 = _step(r, b, u, un, it, use, demand=0)
    assert tail0 is None
    need = (N + 63) // 64 * 64 + (2 * N + 63) // 64 * 64
    assert r._aux_demand == need                         # what the next render will reserve
    l1, t1, g1, tail1 = _step(r, b, u, un, it, use, demand=need)
    assert tail1 is not None and tail1.cap >= need and tail1.used == tail1.cap and tail1.pctx is None
    assert t0 == t1                                      # the forward values: same kernels' arithmetic, bit for bit
    assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0))
    _close(g0, g1)
    # steady state: the demand the second step measured is the same again
    assert r._aux_demand == need


def test_tail_overflow_and_unclaimed_rows():
    c = load_case("trained_deform")
    b, u, un = _batch(c)
    it = int(c["meta/iter_step"])
    r = renderer_for_case(c)
    r.engine.deterministic = True
    use = ("render", "eod", "sn")
    _, t0, g0, _ = _step(r, b, u, un, it, use, demand=0)
    # room for errorondepth's points only: surface_neighbour_error overflows into a stand-alone evaluation
    _, t1, g1, tail = _step(r, b, u, un, it, use, demand=64)
    assert tail is not None and t0 == t1
    _close(g0, g1)
    # far more room than anybody claims: the unclaimed rows are evaluated (zero adjoints) before the backward
    _, t2, g2, tail = _step(r, b, u, un, it, use, demand=1024)
    assert tail is not None and tail.cap >= 1024 and t0 == t2
    _close(g0, g2)
    for k, g in g2.items():
        assert bool(torch.isfinite(g).all()), k
    # an absurd demand (a caller that evaluated a million points after the last render) is capped: a tail row costs a workspace row
    _, t3, g3, tail = _step(r, b, u, un, it, use, demand=1 << 20)
    assert tail is not None and tail.cap <= 4096 + 128 and t0 == t3
    _close(g0, g3)


def test_tail_second_render_before_backward():
    """Two grad-enabled renders before one backward: the later calls go to the LATEST render's tail; both renders back-propagate."""
    c = load_case("trained_deform")
    b, u, un = _batch(c)
    it = int(c["meta/iter_step"])
    res = []
    for demand in (0, 256):
        r = renderer_for_case(c)
        r.engine.deterministic = True
        r.perturb = u is not None
        r._aux_demand = demand
        ret_a = r(b["rays"], iter_step=it, u_perturb=u)
        r._aux_demand = demand
        ret_b = r(b["rays"], iter_step=it, u_perturb=u)
        sdf_loss, angle_loss, _ = r.errorondepth(b["rays"], d_gt=b["depth"], mask=b["mask"], iter_step=it)
        loss = ret_a["color_map"].sum() + ret_b["depth_map"].sum() + sdf_loss + angle_loss
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss), _grads(r)))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * max(1.0, abs(res[0][0]))
    _close(res[0][1], res[1][1])


def test_aux_loss_kernels_match_torch():
    """es_eod_loss / es_sn_loss and their backward against the reference's op chains in torch (endosurf.py:302-317, :334-339)."""
    from endosurf_amd.renderer import _EodLossFn, _SnLossFn
    c = load_case("trained_deform")
    r = renderer_for_case(c)
    eng = r.engine
    g = torch.Generator(device="cuda").manual_seed(3)
    N = 777
    rays = torch.randn(N, 9, generator=g, device="cuda")
    pts = torch.randn(N, 3, generator=g, device="cuda") * 0.7
    mask = (torch.rand(N, 1, generator=g, device="cuda") > 0.3).float()
    sdf = torch.randn(N, 1, generator=g, device="cuda", requires_grad=True)
    go = torch.randn(N, 3, generator=g, device="cuda", requires_grad=True)
    a, bb, inside = _EodLossFn.apply(sdf, go, eng, rays, pts, mask)
    (2.0 * a + 3.0 * bb).backward()
    got = (float(a), float(bb), sdf.grad.clone(), go.grad.clone(), inside.clone())
    sdf.grad = go.grad = None
    ins = (pts.norm(dim=-1, keepdim=True) < 1.0).float() * mask
    den = ins.sum() + 1e-6
    ra, rb = (ins * sdf).abs().sum() / den, torch.relu((rays[:, 3:6] * go).sum(-1, keepdim=True)).abs().sum() / den
    (2.0 * ra + 3.0 * rb).backward()
    assert abs(got[0] - float(ra)) < 1e-5 * abs(float(ra)) and abs(got[1] - float(rb)) < 1e-5 * abs(float(rb))
    assert torch.equal(got[4], ins)
    assert float((got[2] - sdf.grad).abs().max()) < 1e-6 * float(sdf.grad.abs().max())
    assert float((got[3] - go.grad).abs().max()) < 1e-6 * float(go.grad.abs().max())
    # surface-neighbour term, incl. a zero gradient row (torch's norm backward gives 0 there) and the no-valid-ray case
    gg = torch.randn(2 * N, 3, generator=g, device="cuda")
    gg[5] = 0.0
    gg.requires_grad_(True)
    valid = torch.rand(N, generator=g, device="cuda") > 0.4
    valid[5] = True
    out = _SnLossFn.apply(gg, eng, valid)
    (1.7 * out).backward()
    got = (float(out), gg.grad.clone())
    gg.grad = None
    nrm = gg / (gg.norm(dim=-1, keepdim=True) + 1e-10)
    ref = ((nrm[:N] - nrm[N:]).abs() * valid[:, None].float()).sum() / torch.clamp(valid.sum() * 3, min=1).float()
    (1.7 * ref).backward()
    assert abs(got[0] - float(ref)) < 1e-5 * abs(float(ref))
    assert float((got[1] - gg.grad).abs().max()) < 2e-6 * float(gg.grad.abs().max())
    none = _SnLossFn.apply(gg.detach(), eng, torch.zeros(N, dtype=torch.bool, device="cuda"))
    assert float(none) == 0.0


def test_tail_separate_backward_passes():
    """The reference's graphs of the three calls are disjoint: ``loss_a.backward()`` then ``loss_b.backward()`` works there without
    retain_graph.  Here the later calls' nodes hang on the render's node (the token), so a second pass re-enters it: the consumed workspace
    is evaluated again, the accumulated ``.grad`` equals one backward of the sum, and no adjoint of the first pass leaks into the second."""
    c = load_case("trained_deform")
    b, u, un = _batch(c)
    it = int(c["meta/iter_step"])
    N = b["rays"].shape[0]
    need = (N + 63) // 64 * 64 + (2 * N + 63) // 64 * 64
    res = []
    for separate in (False, True):
        r = renderer_for_case(c)
        r.engine.deterministic = True
        r.perturb = u is not None
        for p in r.parameters():
            p.grad = None
        r._aux_demand = need
        ret = r(b["rays"], iter_step=it, u_perturb=u)
        sdf_loss, angle_loss, _ = r.errorondepth(b["rays"], d_gt=b["depth"], mask=b["mask"], iter_step=it)
        sn = r.surface_neighbour_error(rays=b["rays"], mask=b["mask"], iter_step=it, neighbour_rad=0.1, u_neigh=un)
        la = ret["color_map"].abs().sum() + ret["gradient_o_error"]
        lb = sdf_loss + 0.1 * angle_loss
        lc = 0.1 * sn
        if separate:
            la.backward(); lb.backward(); lc.backward()
        else:
            (la + lb + lc).backward()
        torch.cuda.synchronize()
        res.append(_grads(r))
    _close(res[0], res[1])


@pytest.mark.parametrize("n", [0, 1, 100])
def test_reference_sequence_edge_batches(n):
    """Empty, single-ray and ragged (not a multiple of the 64-point tile) batches through the three calls, stand-alone and in the tail:
    finite losses, a backward that runs, finite gradients; with an all-zero mask the surface-neighbour term is exactly 0."""
    import weightgen
    from gpu_util import renderer_for
    r = renderer_for(31, "trained", True)
    r.engine.deterministic = True
    dev = "cuda"
    rays = torch.from_numpy(weightgen.make_rays(5, max(n, 1))[:n]).to(dev)
    tg = {k: torch.from_numpy(v[:n]).to(dev) for k, v in weightgen.make_targets(6, max(n, 1)).items()}
    got = []
    for rep in range(2):                      # second pass: the calls' points go into the render's tail (when there are any)
        for p in r.parameters():
            p.grad = None
        ret = r(rays, iter_step=3, perturb_overwrite=False)
        a, b_, valid = r.errorondepth(rays, d_gt=tg["depth"], mask=tg["mask"], iter_step=3)
        sn = r.surface_neighbour_error(rays=rays, mask=tg["mask"], iter_step=3, neighbour_rad=0.1)
        sn0 = r.surface_neighbour_error(rays=rays, mask=torch.zeros_like(tg["mask"]), iter_step=3, neighbour_rad=0.1)
        assert float(sn0) == 0.0 and tuple(valid.shape) == (n, 1)
        loss = ret["color_map"].sum() + ret["gradient_o_error"] + a + b_ + sn + sn0
        assert bool(torch.isfinite(loss)), float(loss)
        loss.backward()
        torch.cuda.synchronize()
        for k, p in r.named_parameters():
            assert p.grad is None or bool(torch.isfinite(p.grad).all()), k
        got.append((float(a), float(b_), float(loss)))
    if n >= 1:
        assert r._live_tail is not None and r._live_tail[0].used > 0
        assert got[0][:2] == got[1][:2]


def test_deferred_errorondepth_semantics():
    """``errorondepth`` into a live render's tail returns lazily evaluated results (renderer._Lazy / _PendingEod): whatever touches them
    first issues the launches.  Read at once, read after ``surface_neighbour_error`` (the reference's order: ONE launch chain for both
    calls' points), never read, used directly as the loss: same values bit for bit, same gradients as the eager evaluation."""
    from endosurf_amd.renderer import _Lazy
    c = load_case("trained_deform")
    b, u, un = _batch(c)
    it = int(c["meta/iter_step"])
    N = b["rays"].shape[0]
    need = (N + 63) // 64 * 64 + (2 * N + 63) // 64 * 64

    def run(mode, defer=True):
        r = renderer_for_case(c)
        r.render_cfg["defer_errorondepth"] = defer
        r.engine.deterministic = True
        r.perturb = u is not None
        r._aux_demand = need
        ret = r(b["rays"], iter_step=it, u_perturb=u)
        tail = r._live_tail[0]
        a, bb, valid = r.errorondepth(b["rays"], d_gt=b["depth"], mask=b["mask"], iter_step=it)
        assert isinstance(a, _Lazy) == defer and a.requires_grad and tuple(a.shape) == () and a.dtype == torch.float32      # (getters do not force)
        assert (tail.pending is not None) == defer
        early = None
        if mode == "read_at_once":
            assert a.dim() == 0 and a.numel() == 1 and tail.pending is not None      # (metadata methods do not force either)
            alias = a.data                                                    # a getter that hands out the storage: forces
            assert tail.pending is None and not isinstance(alias, _Lazy)
            early = (a.item(), float(bb))
            assert float(alias) == early[0]
        sn = r.surface_neighbour_error(rays=b["rays"], mask=b["mask"], iter_step=it, neighbour_rad=0.1, u_neigh=un)
        if mode != "never_read":
            assert tail.pending is None                                       # sn's evaluation took errorondepth's rows along
        base = ret["color_map"].sum() + (ret["depth_map"] * valid).sum() + 0.1 * sn
        if mode == "never_read":
            loss = base
        elif mode == "loss_is_lazy":
            loss = a                                                          # Tensor.backward on the lazy tensor itself
        else:
            loss = base + a + 0.1 * bb
        vals = (float(a.detach().cpu().numpy()), bb.item()) if mode != "never_read" else None
        loss.backward()
        torch.cuda.synchronize()
        if early is not None:
            assert early == vals
        return vals, _grads(r), type(torch.stack([a, bb]))

    ref_vals, ref_g, _ = run("reference_order", defer=False)
    for mode in ("reference_order", "read_at_once"):
        vals, g, stacked = run(mode)
        assert vals == ref_vals and stacked is torch.Tensor
        _close(ref_g, g, tol=1e-6)
    # never read: the render's backward defines the rows; the gradient is that of the loss without the two terms
    _, g_never, _ = run("never_read")
    _, g_never_ref, _ = run("never_read", defer=False)
    _close(g_never_ref, g_never, tol=1e-6)
    _, g_lazy, _ = run("loss_is_lazy")
    _, g_lazy_ref, _ = run("loss_is_lazy", defer=False)
    _close(g_lazy_ref, g_lazy, tol=1e-6)
