"""The launches that keep a training step inside the library (csrc/step.hip, ABI v5): uniform draws, the step arena, the render
epilogue, the appended auxiliary adjoints and the loss node's seed handling -- through the C ABI and through the Trainer."""
import ctypes as C

import numpy as np
import pytest
import torch

from gpu_util import renderer_for
from philox_ref import uniform as philox_uniform

pytestmark = pytest.mark.gpu


def test_uniform_matches_philox_reference_and_is_uniform():
    from endosurf_amd import _lib
    r = renderer_for(5, "init", True)
    eng = r.engine
    n, seed, sub = 37, 0x1234567890ABCDEF, 5
    out = torch.empty(n, device="cuda")
    _lib.check(eng.lib.es_uniform(_lib.ptr(out), n, seed, sub, None, eng.st()), "es_uniform")
    ref = np.array(philox_uniform(n, seed, sub), np.float32)
    assert np.array_equal(out.cpu().numpy(), ref)                       # bit-exact: integer arithmetic + an exact scaling
    # a device-resident step counter is added to the subsequence (a captured step draws new numbers at every replay)
    step = torch.tensor([3.0, 0.0], dtype=torch.float64, device="cuda")
    _lib.check(eng.lib.es_uniform(_lib.ptr(out), n, seed, sub - 3, _lib.ptr(step), eng.st()), "es_uniform")
    assert np.array_equal(out.cpu().numpy(), ref)
    big = eng.uniform(1 << 20).cpu().numpy()
    assert big.min() >= 0.0 and big.max() < 1.0 and abs(big.mean() - 0.5) < 2e-3 and abs(big.var() - 1 / 12) < 1e-3
    h = np.histogram(big, bins=64, range=(0, 1))[0]
    assert h.min() > 0.9 * big.size / 64 and h.max() < 1.1 * big.size / 64
    torch.manual_seed(7)
    a = eng.uniform(100).cpu()
    b = eng.uniform(100).cpu()
    assert not torch.equal(a, b)                                        # one subsequence per call
    torch.manual_seed(8)
    assert not torch.equal(eng.uniform(100).cpu(), b)


def test_zero_and_scale():
    from endosurf_amd import _lib
    r = renderer_for(5, "init", True)
    eng = r.engine
    x = torch.randn(1000, device="cuda")
    s = torch.tensor([2.5], device="cuda")
    y = torch.empty_like(x)
    _lib.check(eng.lib.es_scale(_lib.ptr(y), _lib.ptr(x), 1000, _lib.ptr(s), eng.st()), "es_scale")
    assert torch.equal(y, x * 2.5)
    _lib.check(eng.lib.es_zero(_lib.ptr(x), 4 * 1000, eng.st()), "es_zero")
    assert float(x.abs().max()) == 0.0


def _batch(n, dev="cuda"):
    from endosurf_amd.trainer import SyntheticScene
    return SyntheticScene(dev, seed=3).batch(n)


def test_step_arena_hands_out_zeroed_slices_and_trains_like_plain_zeros():
    """A Trainer step inside the arena (one memset) equals the same step with every zero-initialised buffer allocated by torch.zeros."""
    from endosurf_amd.trainer import Trainer
    res = []
    for arena in (True, False):
        torch.manual_seed(0)
        r = renderer_for(11, "init", True)
        r.engine.deterministic = True
        tr = Trainer(r)
        if not arena:
            r.engine.arena_begin = lambda extra_floats=0: None            # zeros() then falls through to torch.zeros
        b = _batch(256)
        u, un = torch.rand(256, 1, device="cuda", generator=torch.Generator("cuda").manual_seed(1)), torch.rand(256, 3, device="cuda", generator=torch.Generator("cuda").manual_seed(2))
        for it in range(3):
            tr.update_learning_rate(it + 1)
            loss, _, _ = tr.train_step(b, it + 1, u_perturb=u, u_neigh=un)
        res.append((float(loss), r.model._flat.detach().clone()))
        if arena:
            assert r.engine._arena is not None and r.engine._arena_off > r.engine.n_weff + r.engine.n_param and not r.engine._arena_on
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])


def test_loss_node_scales_its_adjoints_for_a_foreign_seed():
    """loss.backward() with the engine's ones hands the adjoints on unscaled; any other seed (here 2 x loss) goes through es_scale."""
    from endosurf_amd.trainer import compute_loss_fused
    grads = []
    for scale in (None, 2.0):
        torch.manual_seed(0)
        r = renderer_for(11, "init", True)
        r.engine.deterministic = True
        b = _batch(128)
        u, un = torch.full((128, 1), 0.3, device="cuda"), torch.full((128, 3), 0.6, device="cuda")
        total, _, _ = compute_loss_fused(r, b, 1, u_perturb=u, u_neigh=un)
        if scale is None:
            total.backward(gradient=r.engine.ones1.reshape(total.shape))
        else:
            (total * scale).backward()
        grads.append(dict(r.named_parameters())["model.sdf_network.net.2.weight_v"].grad.detach().clone())
    rel = float((grads[1] - 2.0 * grads[0]).norm() / (2.0 * grads[0]).norm())
    assert rel < 1e-6, rel


def test_drawn_step_is_reproducible_from_the_torch_seed():
    """Without explicit draws a step takes its uniform numbers from es_uniform keyed by torch's seed: same seed, same trajectory."""
    from endosurf_amd.trainer import Trainer
    out = []
    for _ in range(2):
        torch.manual_seed(123)
        r = renderer_for(11, "init", True)
        r.engine.deterministic = True
        tr = Trainer(r)
        b = _batch(128)
        for it in range(2):
            loss, _, _ = tr.train_step(b, it + 1)
        out.append(float(loss))
    assert out[0] == out[1]


def test_param_grads_between_steps_and_graph_rng_space():
    """What a user of Trainer.train_step may rely on ("gradients of a Trainer step"): after a step every parameter's
    ``.grad`` is a view of the step's flat gradient (valid, finite, non-zero) until the NEXT step begins; a view kept across steps is
    overwritten then (the step arena is one block, cleared and refilled by every step), so gradients that must survive are cloned.
    And the random draws of graph-mode steps come from a subsequence space disjoint from the eager steps' (ADVICE r4)."""
    from endosurf_amd.trainer import SyntheticScene, Trainer
    r = renderer_for(5, "trained", True)
    tr = Trainer(r, lr=1e-3)
    sc = SyntheticScene("cuda", seed=3)
    tr.update_learning_rate(100)
    tr.train_step(sc.batch(128), 100)
    named = dict(r.named_parameters())
    p = named["model.sdf_network.net.3.weight_v"]
    flat = tr.optimizer.flat_grad()
    off = r.model._layout["sdf_network.net.3.weight_v"][0]
    assert p.grad is not None and p.grad.data_ptr() == flat.data_ptr() + 4 * off          # a view of the step's flat buffer
    g1 = p.grad.clone()
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    kept = p.grad
    tr.train_step(sc.batch(128), 101)
    assert not torch.equal(kept, g1)                      # the retained view was overwritten by the next step ...
    assert torch.equal(kept, p.grad) or p.grad.data_ptr() != kept.data_ptr()       # ... (it is that step's gradient, or a dead slice)
    # eager draws: subsequence = call index; graph-mode draws: 2^63 + device step counter -- never the same stream
    eng = r.engine
    torch.manual_seed(11)
    eng._rng_calls = 4
    step = torch.tensor([4.0, 0.0], dtype=torch.float64, device="cuda")
    a = eng.uniform(64).cpu()
    b = eng.uniform(64, step).cpu()
    assert not torch.equal(a, b)
    seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
    assert np.array_equal(b.numpy(), np.array(philox_uniform(64, seed, (1 << 63) + 4), np.float32))
