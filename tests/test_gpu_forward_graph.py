"""``renderer(rays)`` under no_grad (the reference's eval loop, trainer_endosurf.py:221-240) as one captured hipGraph:
EndoSurfRenderer._forward_captured.  The second call of a (ray count, sampling mode, weights) key captures; replays must return what the
eager launches return, bit for bit, in tensors the caller owns."""
import pytest
import torch

from gpu_util import renderer_for
from oracle_util import RENDER_CFG

pytestmark = pytest.mark.gpu
KEYS = ("color_map", "depth_map", "gradients_o", "gradient_o_error", "weights", "weight_max", "cdf", "s_val")


def _rays(n, seed=3):
    from endosurf_amd.trainer import SyntheticScene
    return SyntheticScene("cuda", seed=seed).batch(n)["rays"]


@pytest.mark.parametrize("use_deform", [True, False])
def test_captured_forward_is_bit_identical_to_the_eager_one(use_deform):
    r = renderer_for(21, "trained", use_deform)
    e = renderer_for(21, "trained", use_deform, render_cfg=dict(RENDER_CFG, forward_graph=False))
    r.engine.deterministic = e.engine.deterministic = True          # (the eikonal batch sums are fp32 atomics otherwise: order-dependent)
    a, b = _rays(256, 3), _rays(256, 4)
    with torch.no_grad():
        first = r(a, iter_step=20000, perturb_overwrite=False)              # eager: first call of the key
        assert r._fwd_graph["graph"] is None
        second = r(b, iter_step=20000, perturb_overwrite=False)             # captures + replays
        assert r._fwd_graph["graph"] is not None
        third = r(a, iter_step=30000, perturb_overwrite=False)              # replay with another cos-anneal ratio (a device scalar)
        graph = r._fwd_graph["graph"]
        ea = e(a, iter_step=20000, perturb_overwrite=False)
        eb = e(b, iter_step=20000, perturb_overwrite=False)
        ea3 = e(a, iter_step=30000, perturb_overwrite=False)
        assert e._fwd_graph is None
    assert set(first) == set(second) == set(ea) == set(KEYS)
    for k in KEYS:
        assert second[k].shape == eb[k].shape and second[k].dtype == eb[k].dtype, k
        assert torch.equal(first[k], ea[k]), k
        assert torch.equal(second[k], eb[k]), k
        assert torch.equal(third[k], ea3[k]), k
    assert r._fwd_graph["graph"] is graph
    # the outputs belong to the caller: a later replay does not touch them
    keep = {k: v.clone() for k, v in second.items()}
    with torch.no_grad():
        r(a, iter_step=20000, perturb_overwrite=False)
    for k in KEYS:
        assert torch.equal(second[k], keep[k]), k


def test_captured_forward_draws_new_jitter_and_follows_the_weights():
    r = renderer_for(21, "trained", True)
    a = _rays(128)
    with torch.no_grad():
        outs = [r(a, iter_step=100) for _ in range(4)]                       # perturb = True: torch.rand inside the graph
    assert r._fwd_graph["graph"] is not None
    depths = {float(o["depth_map"].double().sum()) for o in outs}
    assert len(depths) == 4, depths                                          # every replay draws its own stratified jitter
    # new weights: the key changes, the next call runs eagerly, the one after it re-captures -- and renders the new weights
    g0 = r._fwd_graph["graph"]
    with torch.no_grad():
        ref = r(a, iter_step=100, perturb_overwrite=False)
        r(a, iter_step=100, perturb_overwrite=False)
        for p in r.parameters():
            p.mul_(1.01)
        x = r(a, iter_step=100, perturb_overwrite=False)
        assert r._fwd_graph["graph"] is None
        y = r(a, iter_step=100, perturb_overwrite=False)
        assert r._fwd_graph["graph"] is not None and r._fwd_graph["graph"] is not g0
    assert torch.equal(x["color_map"], y["color_map"]) and not torch.equal(x["color_map"], ref["color_map"])


def test_captured_forward_stands_down():
    """Foreign keyword arguments, grad mode and the per-kernel timers keep the eager path."""
    r = renderer_for(21, "trained", True)
    a = _rays(64)
    u = torch.rand(64, 1, device="cuda")
    with torch.no_grad():
        r(a, iter_step=1, u_perturb=u); r(a, iter_step=1, u_perturb=u)
    assert r._fwd_graph is None
    r(a, iter_step=1); r(a, iter_step=1)                                      # grad enabled
    assert r._fwd_graph is None
    r.engine.timing_enable(True)
    try:
        with torch.no_grad():
            r(a, iter_step=1); r(a, iter_step=1)
        assert r._fwd_graph is None
    finally:
        r.engine.timing_drain()
        r.engine.timing_enable(False)


def test_captured_forward_keeps_a_slot_per_shape():
    """The eval loop's pattern (trainer_endosurf.py:221-240): full chunks, one shorter last chunk, next image: both shapes stay captured
    (at most three slots); a grad-enabled render releases them."""
    r = renderer_for(21, "trained", True)
    full, last = _rays(256, 5), _rays(96, 6)
    with torch.no_grad():
        for _ in range(2):
            r(full, iter_step=7, perturb_overwrite=False)
            r(last, iter_step=7, perturb_overwrite=False)
        slots = r.__dict__["_fwd_graphs"]
        assert len(slots) == 2 and all(s["graph"] is not None for s in slots.values())
        graphs = [s["graph"] for s in slots.values()]
        a = r(full, iter_step=7, perturb_overwrite=False)
        b = r(last, iter_step=7, perturb_overwrite=False)
        assert [s["graph"] for s in r.__dict__["_fwd_graphs"].values()] == graphs          # replays, no re-capture
        for n in (32, 64, 128):                                                           # three more shapes: the oldest slots go
            r(_rays(n), iter_step=7, perturb_overwrite=False)
        assert len(r.__dict__["_fwd_graphs"]) == 3
    assert a["color_map"].shape[0] == 256 and b["color_map"].shape[0] == 96
    r(full, iter_step=7)["color_map"].sum().backward()                                    # grad-enabled render: the slots are released
    assert r._fwd_graph is None
