"""GPU parity of the hand-written backward: parameter gradients of the fused point evaluation and of the full
renderer / training loss against autograd on the fp64 oracle."""
import json
import os

import numpy as np
import pytest
import torch

import weightgen
from gpu_util import renderer_for, renderer_for_case
from oracle import endosurf_oracle as O
from oracle_util import CASES, RENDER_CFG, T, load_case, oracle_for

pytestmark = pytest.mark.gpu
LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _oracle(seed, mode, use_deform, dtype):
    state = weightgen.make_state(seed, mode, use_deform)
    params = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in state.items()}
    return O.OracleNet(params, use_deform), params


def _grad_table(r, params):
    """relative L2 error per parameter tensor: |g_hip - g_ref| / (|g_ref| + tiny)"""
    rows = {}
    named = dict(r.named_parameters())
    for k, p in params.items():
        ref = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().double().reshape(-1)
        got = named["model." + k].grad
        got = (got if got is not None else torch.zeros_like(named["model." + k])).detach().double().cpu().reshape(-1)
        rows[k] = (float((got - ref).norm() / (ref.norm() + 1e-30)), float(ref.norm()), float(got.norm()))
    return rows


def _dump(name, rows):
    os.makedirs(LOG, exist_ok=True)
    with open(os.path.join(LOG, f"grad_{name}.json"), "w") as f:
        json.dump(rows, f, indent=1)


# (per-tensor bound, median bound) on the relative L2 error of a parameter-gradient tensor of test_point_backward; "split": the same test
# through the split-precision chain kernels (tests/test_gpu_train_x3.py), at that family's own measured level
POINT_TOL = {"fp32": (5e-4, 1e-4), "split": (5e-4, 1e-4)}


@pytest.mark.parametrize("mode,use_deform,color", [("init", True, True), ("trained", True, True), ("trained", False, True),
                                                   ("trained", True, False)])
def test_point_backward(mode, use_deform, color):
    M = 200
    seed = 41
    r = renderer_for(seed, mode, use_deform)
    net, params = _oracle(seed, mode, use_deform, torch.float64)
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.uniform(-0.7, 0.7, size=(M, 3)).astype(np.float32))
    d = rng.normal(size=(M, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = torch.from_numpy(d.astype(np.float32))
    t = torch.from_numpy(rng.uniform(size=(M,)).astype(np.float32))
    ws, wg, wc = (torch.from_numpy(rng.normal(size=s).astype(np.float32)) for s in ((M, 1), (M, 3), (M, 3)))

    from endosurf_amd import _lib
    from endosurf_amd.renderer import _PointEvalFn
    weff, packed = r._weights()
    flags = r._flags(weff) | (_lib.PF_COLOR if color else 0)
    pts = r.engine.points(x=x.cuda(), t=t.cuda(), dirs=d.cuda())
    outs = _PointEvalFn.apply(weff, packed, r.engine, pts, flags)
    loss = (outs[0] * ws.cuda()).sum() + (outs[1] * wg.cuda()).sum()
    if color:
        loss = loss + (outs[2] * wc.cuda()).sum()
    loss.backward()
    torch.cuda.synchronize()

    pe = net.point_eval(x.double(), d.double(), t.double()[:, None], with_color=color)
    ref = (pe["sdf"] * ws.double()).sum() + (pe["g_o"] * wg.double()).sum()
    if color:
        ref = ref + (pe["rgb"] * wc.double()).sum()
    ref.backward()
    assert abs(float(loss) - float(ref)) < 2e-3 * max(1.0, abs(float(ref)))
    rows = _grad_table(r, params)
    split = bool(r.engine.split_precision)
    _dump(f"point_{mode}_{int(use_deform)}_{int(color)}" + ("_split" if split else ""), rows)
    # At fixed points the kernels agree with autograd on the fp64 oracle to 2.3e-6 ... 2.3e-5 relative L2 per parameter tensor
    # (82 tensors x 4 cases, gpurun_out/grad_point_*.json of round 4).  The gate sits at max(20 x measured, 5e-4) = 5e-4 per tensor and
    # 1e-4 for the median (round 5; it was 2e-2 / 2e-3, which a 1 % defect in one small tensor would have passed).  No tensor needs more.
    per_tensor, median = POINT_TOL["split" if split else "fp32"]
    bad = {k: v for k, v in rows.items() if v[0] > per_tensor and v[1] > 1e-7 and k != "deviation_network.variance"}
    assert not bad, bad
    med = float(np.median([v[0] for k, v in rows.items() if v[1] > 1e-7]))
    assert med < median, med


def _render_scalar(ret, c, dt, dev):
    cw, dw, gw, ww = (torch.tensor(c[f"scal/{k}"], dtype=dt, device=dev) for k in ("cw", "dw", "gw", "ww"))
    return ((ret["color_map"] * cw).sum() + (ret["depth_map"] * dw).sum() + (ret["gradients_o"] * gw).sum()
            + (ret["weights"] * ww).sum() + 0.5 * ret["gradient_o_error"] + (ret["cdf"] * ww).sum() * 0.1
            + ret["s_val"].sum() * 0.01)


FLOOR = 5e-3
WIDER = {}          # (case, tensor) -> budget: none needed (round 4).  With the floor at 5e-3 (was 2e-2) the only tensors above it were
                    # init_deform's colour layer 6 (bias, weight_v: 0.694 %), and the reference's OWN fp32-vs-fp64 error on those tensors is
                    # 0.695 % (goldens: scalgraderr/*/rel, whole tensor) -- the 32 sampled entries had put it at 0.015 % / 0.19 %


def _check_rows(rows, c, prefix64, prefix32, name, r=None):
    """Per parameter tensor, against the fp64 reference / oracle:
      (1) norm within 3x the reference's own fp32-vs-fp64 norm difference (golden norm summaries);
      (2) DIRECTION: relative L2 error of the whole tensor vs autograd on the fp64 oracle <= max(3x the reference's own
          fp32-vs-fp64 relative error, FLOOR = 5e-3; round 4: was 2e-2) -- the reference's own error is the larger of its whole-tensor
          figure (goldens ``{scalgrad,grad}err/<tensor>/rel``: the reference run in both precisions by tools/make_golden.py) and the
          estimate from the 32 sampled entries per tensor the goldens hold for both precisions (grad64/*/val vs grad/*/val);
      (3) the same 32 sampled entries of the HIP gradient against the reference's fp64 values, same budget."""
    bad, table = {}, {}
    named = dict(r.named_parameters()) if r is not None else {}
    for k, (rel, nref, ngot) in rows.items():
        if nref < 1e-9:
            continue
        n64, n32 = float(c[f"{prefix64}/{k}/norm"]), float(c[f"{prefix32}/{k}/norm"])
        ref_noise = abs(n32 - n64) / (n64 + 1e-30)
        tol = max(5e-3, 3 * ref_noise + 2e-2 * (ref_noise > 1e-3))
        if abs(ngot - n64) / (n64 + 1e-30) > tol:
            bad[k] = ("norm", rel, ngot, n64, n32)
            continue
        v64 = np.asarray(c[f"{prefix64}/{k}/val"], np.float64)
        v32 = np.asarray(c[f"{prefix32}/{k}/val"], np.float64)
        idx = np.asarray(c[f"{prefix64}/{k}/idx"])
        samp_norm = np.linalg.norm(v64)
        # the reference's own relative error, from the sampled entries (scaled to the tensor: the samples' share of the norm varies)
        ref_rel = np.linalg.norm(v32 - v64) / (samp_norm + 1e-30) if samp_norm > 1e-3 * n64 * np.sqrt(len(idx) / max(len(idx), 1)) else 0.0
        ekey = f"{prefix32}err/{k}/rel"
        if ekey in c:
            ref_rel = max(ref_rel, float(c[ekey]))
        budget = max(3 * ref_rel, WIDER.get((name, k), FLOOR))
        table[k] = (rel, budget, ref_rel)
        if rel > budget:
            bad[k] = ("direction", rel, budget, ref_rel)
            continue
        if r is not None and k != "deviation_network.variance":
            got = named["model." + k].grad.detach().double().cpu().reshape(-1).numpy()[idx]
            # sampled entries: compare on the scale of the tensor's RMS entry (single entries can be ~0)
            rms = n64 / np.sqrt(max(named["model." + k].numel(), 1))
            err = np.linalg.norm(got - v64) / (np.linalg.norm(v64) + np.sqrt(len(idx)) * rms)
            if err > budget:
                bad[k] = ("samples", err, budget, ref_rel)
    _dump(f"budget_{prefix64}_{name}" + ("_split" if os.environ.get("ES_SPLIT_BF16", "0") not in ("0", "") else ""),
          {k: v for k, v in table.items()})
    assert not bad, (name, bad)


@pytest.mark.parametrize("name", CASES)
def test_render_scalar_param_grads(name):
    c = load_case(name)
    r = renderer_for_case(c)
    rays = torch.from_numpy(c["rays"]).cuda()
    u = torch.from_numpy(c["u_perturb"]).cuda() if "u_perturb" in c else None
    it = int(c["meta/iter_step"])
    ret = r(rays, iter_step=it, perturb_overwrite=u is not None, u_perturb=u)
    scal = _render_scalar(ret, c, torch.float32, "cuda")
    scal.backward()
    torch.cuda.synchronize()
    R, params = oracle_for(c, torch.float64, requires_grad=True)
    ref = _render_scalar(R.render_rays(T(c["rays"], torch.float64), it, None if u is None else T(c["u_perturb"], torch.float64)), c,
                         torch.float64, "cpu")
    ref.backward()
    v64 = float(c["scal64/value"])
    assert abs(float(scal) - v64) < 3 * abs(float(c["scal/value"]) - v64) + 1e-4 * max(1.0, abs(v64))
    rows = _grad_table(r, params)
    _dump(f"scal_{name}", rows)
    _check_rows(rows, c, "scalgrad64", "scalgrad", name, r)


@pytest.mark.parametrize("name", CASES)
def test_training_loss_param_grads(name):
    from endosurf_amd.trainer import compute_loss
    c = load_case(name)
    r = renderer_for_case(c)
    dev = "cuda"
    batch = dict(rays=torch.from_numpy(c["rays"]).to(dev), color=torch.from_numpy(c["target/color"]).to(dev),
                 depth=torch.from_numpy(c["target/depth"]).to(dev), mask=torch.from_numpy(c["target/mask"]).to(dev),
                 color_mask=torch.from_numpy(c["target/color_mask"]).to(dev))
    u = torch.from_numpy(c["u_perturb"]).to(dev) if "u_perturb" in c else None
    r.perturb = u is not None
    total, terms, _ = compute_loss(r, batch, int(c["meta/iter_step"]), u_perturb=u, u_neigh=torch.from_numpy(c["u_neigh"]).to(dev))
    total.backward()
    torch.cuda.synchronize()
    for k, v in terms.items():
        v64, v32 = float(c[f"loss64/{k}"]), float(c[f"loss/{k}"])
        assert abs(float(v) - v64) < 3 * abs(v32 - v64) + 2e-5 * max(1.0, abs(v64)), (k, float(v), v64, v32)
    t64 = float(c["loss64/total"])
    assert abs(float(total) - t64) < 3 * abs(float(c["loss/total"]) - t64) + 5e-5 * max(1.0, abs(t64))
    R, params = oracle_for(c, torch.float64, requires_grad=True)
    dt = torch.float64
    ob = {k: T(c[f"target/{k}"], dt) for k in ("color", "depth", "mask", "color_mask")}
    ob["rays"] = T(c["rays"], dt)
    ototal, _, _ = O.train_loss(R, ob, int(c["meta/iter_step"]), None if u is None else T(c["u_perturb"], dt), T(c["u_neigh"], dt))
    ototal.backward()
    rows = _grad_table(r, params)
    _dump(f"train_{name}", rows)
    _check_rows(rows, c, "grad64", "grad", name, r)


@pytest.mark.parametrize("name", ["trained_deform", "trained_nodeform"])
def test_fused_training_loss_matches_unfused(name):
    """compute_loss_fused (aux points inside the render launches, fused loss kernel) == compute_loss (reference call sequence)."""
    from endosurf_amd.trainer import compute_loss, compute_loss_fused
    c = load_case(name)
    dev = "cuda"
    batch = dict(rays=torch.from_numpy(c["rays"]).to(dev), color=torch.from_numpy(c["target/color"]).to(dev),
                 depth=torch.from_numpy(c["target/depth"]).to(dev), mask=torch.from_numpy(c["target/mask"]).to(dev),
                 color_mask=torch.from_numpy(c["target/color_mask"]).to(dev))
    u = torch.from_numpy(c["u_perturb"]).to(dev) if "u_perturb" in c else None
    un = torch.from_numpy(c["u_neigh"]).to(dev)
    res = []
    for fn in (compute_loss, compute_loss_fused):
        r = renderer_for_case(c)
        r.perturb = u is not None
        total, terms, _ = fn(r, batch, int(c["meta/iter_step"]), u_perturb=u, u_neigh=un)
        total.backward()
        torch.cuda.synchronize()
        res.append((float(total), {k: float(v) for k, v in terms.items()}, {k: p.grad.clone() for k, p in r.named_parameters()}))
    (t0, terms0, g0), (t1, terms1, g1) = res[:2]
    for tb, termsb, gb in res[1:]:
        assert abs(t0 - tb) < 1e-5 * max(1.0, abs(t0)), (t0, tb)
        for k in terms0:
            assert abs(terms0[k] - termsb[k]) < 1e-5 * max(1.0, abs(terms0[k])), k
        for k in g0:
            n = float(g0[k].norm())
            assert float((g0[k] - gb[k]).norm()) <= 2e-4 * n + 1e-7, (k, float((g0[k] - gb[k]).norm()), n)
    t64 = float(c["loss64/total"])
    assert abs(t1 - t64) < 3 * abs(float(c["loss/total"]) - t64) + 5e-5 * max(1.0, abs(t64))
