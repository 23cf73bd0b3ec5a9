"""(1) Bounded training workspace: a grad-enabled render larger than ``workspace_gb`` is chunked over rays with re-evaluation in
the backward and gives the same outputs and parameter gradients.  (2) Optimiser-state checkpoints in
torch.optim.Adam's own format, the one the reference stores as ckpt["optimizer"] (trainer_endosurf.py:76-92)."""
import copy

import numpy as np
import pytest
import torch

from gpu_util import renderer_for

pytestmark = pytest.mark.gpu


def _batch(n, seed=3):
    from endosurf_amd.trainer import SyntheticScene
    return SyntheticScene("cuda", seed=seed).batch(n)


def _render_loss(r, rays, u):
    ret = r(rays, iter_step=2000, u_perturb=u)
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    w = lambda t: torch.randn(t.shape, device="cuda", generator=g)
    loss = ((ret["color_map"] * w(ret["color_map"])).sum() + (ret["depth_map"] * w(ret["depth_map"])).sum() + 0.3 * ret["gradient_o_error"]
            + 0.01 * (ret["gradients_o"] * w(ret["gradients_o"])).sum() + 0.1 * (ret["weights"] * w(ret["weights"])).sum()
            + 0.1 * (ret["cdf"] * w(ret["cdf"])).sum() + (ret["weight_max"] * w(ret["weight_max"])).sum() + 0.01 * ret["s_val"].sum())
    return loss, ret


def test_chunked_render_matches_unchunked():
    n = 320                                   # 5 chunks of 64 rays
    b = _batch(n)
    u = torch.rand(n, 1, device="cuda")
    res = []
    for gb in (64.0, 64 * 64 * 110e3 / 1e9):  # plenty | room for ~64 rays x 64 samples only
        r = renderer_for(41, "trained", True)
        r.workspace_gb = gb
        r.engine.x3_infer_min = 1             # (split-precision mode, ES_SPLIT_BF16=1: the same kernel family for the 4 096-point chunks
        loss, ret = _render_loss(r, b["rays"], u)       # as for the 20 480-point batch; no effect on the fp32 default)
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss), {k: v.detach().clone() for k, v in ret.items()}, {k: p.grad.clone() for k, p in r.named_parameters()},
                    r._chunk_rays(n, 64, r._flags(r._weights()[0]))))
    (l0, o0, g0, c0), (l1, o1, g1, c1) = res
    assert c0 == 0 and c1 == 64
    assert abs(l0 - l1) < 1e-4 * max(1.0, abs(l0))
    for k in o0:
        assert torch.allclose(o0[k], o1[k], rtol=1e-5, atol=1e-6), k
    for k in g0:
        n0 = float(g0[k].norm())
        assert float((g0[k] - g1[k]).norm()) <= 1e-4 * n0 + 1e-8, (k, float((g0[k] - g1[k]).norm()), n0)


def test_workspace_budget_too_small_raises():
    from endosurf_amd._lib import EndoSurfHipError
    r = renderer_for(41, "trained", True)
    r.workspace_gb = 1e-3
    with pytest.raises(EndoSurfHipError, match="workspace_gb"):
        r(_batch(128)["rays"], iter_step=1)
    with torch.no_grad():                     # no saved activations: no budget applies
        r(_batch(128)["rays"], iter_step=1)


@pytest.mark.parametrize("use_deform", [True, False])
def test_flat_adam_state_dict_is_torch_adams(use_deform):
    """FlatAdam <-> torch.optim.Adam: (a) a state_dict written by torch.optim.Adam over get_train_params() (what the reference
    trainer saves) resumes FlatAdam on the same trajectory; (b) FlatAdam's state_dict loads into torch.optim.Adam."""
    from endosurf_amd.trainer import FlatAdam, Trainer
    b = _batch(256)
    n = 256
    u, un = torch.rand(n, 1, device="cuda"), torch.rand(n, 3, device="cuda")

    def grads(r, tr, it):
        tr.optimizer.zero_grad(set_to_none=True)
        loss, _, _ = tr.loss_fn(r, b, it, tr.loss_weights, tr.surf_neig_rad, u, un)
        loss.backward()

    # reference-style trainer: torch.optim.Adam over the per-tensor parameters, 2 steps, then "checkpoint"
    r_t = renderer_for(52, "trained", use_deform)
    r_t.engine.deterministic = True
    tr_t = Trainer(r_t, flat_adam=False, warm_up_end=1)
    assert isinstance(tr_t.optimizer, torch.optim.Adam)
    for it in (1, 2):
        tr_t.update_learning_rate(it)
        tr_t.train_step(b, it, u_perturb=u, u_neigh=un)
    ckpt = r_t.save_checkpoint()
    ckpt["n_iter"] = 2
    ckpt["optimizer"] = copy.deepcopy(tr_t.optimizer.state_dict())
    keys = set(ckpt["optimizer"]["param_groups"][0])

    # resume with FlatAdam from that checkpoint and take step 3 on both
    r_f = renderer_for(53, "init", use_deform)          # different weights on purpose: everything must come from the checkpoint
    r_f.engine.deterministic = True
    tr_f = Trainer(r_f, warm_up_end=1)
    assert isinstance(tr_f.optimizer, FlatAdam)
    assert tr_f.load_checkpoint(ckpt) == 3
    assert tr_f.optimizer.step_count == 2
    for tr in (tr_t, tr_f):
        tr.update_learning_rate(3)
        tr.train_step(b, 3, u_perturb=u, u_neigh=un)
    torch.cuda.synchronize()
    pt, pf = dict(r_t.named_parameters()), dict(r_f.named_parameters())
    for k in pt:
        d = float((pt[k] - pf[k]).abs().max())
        assert d <= 2e-6 * max(1.0, float(pt[k].abs().max())), (k, d)

    # (b) the other direction: FlatAdam -> torch.optim.Adam
    sd = tr_f.save_checkpoint(3)["optimizer"]
    assert set(sd) == {"state", "param_groups"} and len(sd["param_groups"]) == 1
    assert keys <= set(sd["param_groups"][0]), keys - set(sd["param_groups"][0])
    n_params = len(tr_t.params)
    assert sd["param_groups"][0]["params"] == list(range(n_params)) and sorted(sd["state"]) == list(range(n_params))
    opt = torch.optim.Adam(tr_t.params, lr=5e-4)
    opt.load_state_dict(sd)
    for i, p in enumerate(tr_t.params):
        st = opt.state[p]
        assert tuple(st["exp_avg"].shape) == tuple(p.shape) and int(float(st["step"])) == 3
    ref = tr_t.optimizer.state_dict()["state"]
    for i in (0, 5, n_params - 1):
        assert torch.allclose(ref[i]["exp_avg"], sd["state"][i]["exp_avg"], rtol=1e-4, atol=1e-9)
        assert torch.allclose(ref[i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"], rtol=1e-4, atol=1e-12)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process")
def test_two_renderers_on_two_devices_in_one_process():
    """Per-device constant tables / kernel attributes and per-device streams."""
    r0 = renderer_for(41, "trained", True)
    from endosurf_amd import EndoSurfRenderer
    from gpu_util import net_cfg, state_to_ckpt
    from oracle_util import RENDER_CFG
    import weightgen
    r1 = EndoSurfRenderer(dict(RENDER_CFG), net_cfg(True), device="cuda:1")
    r1.load_checkpoint(state_to_ckpt(weightgen.make_state(41, "trained", True), True))
    rays = _batch(128)["rays"]
    with torch.no_grad():
        a = r0(rays, iter_step=1, perturb_overwrite=False)["color_map"]
        b = r1(rays.to("cuda:1"), iter_step=1, perturb_overwrite=False)["color_map"]
    assert torch.allclose(a.cpu(), b.cpu(), atol=1e-6)
