"""Whole-step hipGraph (Trainer.train_step_graph): the captured training step -- weight-norm packing, ray marching + secant, sampling,
render, loss, backward, Adam, with the learning-rate schedule / bias corrections / cos-anneal ratio computed on the device -- follows the
eager trajectory."""
import math

import numpy as np
import pytest
import torch

from gpu_util import renderer_for

pytestmark = pytest.mark.gpu


def _run(graph, n_steps, n_rays=256, anneal_end=50000):
    from endosurf_amd.trainer import SyntheticScene, Trainer
    from oracle_util import RENDER_CFG
    r = renderer_for(5, "trained", True, render_cfg=dict(RENDER_CFG, anneal_end=anneal_end))
    r.engine.deterministic = True
    tr = Trainer(r, lr=1e-3, n_iter=40, warm_up_end=4, lr_alpha=0.05)
    sc = SyntheticScene("cuda", seed=77)
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    losses = []
    for it in range(1, n_steps + 1):
        b = sc.batch(n_rays)
        b["u_perturb"] = torch.rand(n_rays, 1, device="cuda", generator=gen)
        b["u_neigh"] = torch.rand(n_rays, 3, device="cuda", generator=gen)
        if graph:
            losses.append(float(tr.train_step_graph(b, it)))
        else:
            tr.update_learning_rate(it)
            u, un = b.pop("u_perturb"), b.pop("u_neigh")
            losses.append(float(tr.train_step(b, it, u_perturb=u, u_neigh=un)[0]))
    return np.array(losses), r.model._flat.detach().clone(), tr


def test_graph_step_follows_the_eager_trajectory():
    """8 steps through the warm-up and into the cosine decay (the schedule changes every step; anneal_end = 6 makes the cos-anneal ratio
    move too): the first two calls run eagerly, the third captures, the rest replay."""
    le, pe, _ = _run(False, 8, anneal_end=6)
    lg, pg, tr = _run(True, 8, anneal_end=6)
    assert tr._graph["graph"] is not None and tr.optimizer.step_count == 8
    assert np.allclose(le, lg, rtol=2e-5, atol=1e-6), (le, lg)
    assert float((pe - pg).abs().max()) <= 2e-5 * float(pe.abs().max()), float((pe - pg).abs().max())
    assert float((pe - renderer_for(5, "trained", True).model._flat).abs().max()) > 1e-3           # and the parameters did move
    # the device-side schedule against the host formulas it replaces
    st = tr._graph["state"].cpu().numpy()
    assert st[0] == 8 and st[1] == 8
    from endosurf_amd.trainer import lr_factor
    lr = 1e-3 * lr_factor(8, 40, 4, 0.05)
    want = [lr / (1 - 0.9 ** 8), math.sqrt(1 - 0.999 ** 8), 1.0, 1.0]
    assert np.allclose(tr._graph["scal"].cpu().numpy(), want, rtol=1e-6)


def test_graph_step_random_draws_and_recapture_on_a_new_shape():
    """Without supplied draws the captured step uses torch's graph-safe generator (new jitter at every replay: the loss keeps changing),
    and a batch of another shape captures a second graph."""
    from endosurf_amd.trainer import SyntheticScene, Trainer
    r = renderer_for(5, "trained", True)
    tr = Trainer(r, lr=5e-4, n_iter=100, warm_up_end=10)
    sc = SyntheticScene("cuda", seed=1)
    b = sc.batch(256)
    losses = [float(tr.train_step_graph(b, it)) for it in range(1, 7)]
    assert len(set(np.round(losses, 6))) == 6 and all(np.isfinite(losses))
    g0 = tr._graph["graph"]
    tr.train_step_graph(sc.batch(128), 7)
    assert tr._graph["graph"] is None and tr._graph["key"][0] == (128, 9) and g0 is not None


def test_captured_step_survives_a_regrown_arena():
    """the step arena's address is baked into a captured step (its memset and every launch that uses a slice of it).
    An eager step with a LARGER batch makes the engine allocate a new arena; the graph keeps the old block alive, so later replays
    neither write into recycled memory nor see other tensors' data: the replayed trajectory equals that of a trainer that never left
    the graph."""
    from endosurf_amd.trainer import SyntheticScene, Trainer

    def run(interleave):
        r = renderer_for(5, "trained", True)
        r.engine.deterministic = True
        tr = Trainer(r, lr=1e-3, n_iter=40, warm_up_end=4)
        sc = SyntheticScene("cuda", seed=9)
        gen = torch.Generator(device="cuda"); gen.manual_seed(4)
        small = [sc.batch(128) for _ in range(6)]
        for b in small:
            b["u_perturb"] = torch.rand(128, 1, device="cuda", generator=gen)
            b["u_neigh"] = torch.rand(128, 3, device="cuda", generator=gen)
        big = sc.batch(1024)
        losses = []
        for it, b in enumerate(small, 1):
            losses.append(float(tr.train_step_graph(b, it)))
            if interleave and it == 4:
                assert tr._graph["graph"] is not None
                arena0 = r.engine._arena
                flat0 = r.model._flat.detach().clone()
                opt_state = (tr.optimizer.exp_avg.clone(), tr.optimizer.exp_avg_sq.clone(), tr.optimizer.step_count)
                tr.train_step(big, it)                          # larger batch: the engine needs (and allocates) a larger arena
                assert r.engine._arena is not arena0 and tr._graph["arena"] is arena0
                junk = [torch.full((arena0.numel(),), float("nan"), device="cuda") for _ in range(2)]      # would land in a freed arena
                # undo the eager step's update so that both trajectories see the same parameters / moments
                with torch.no_grad():
                    r.model._flat.copy_(flat0)
                tr.optimizer.exp_avg.copy_(opt_state[0]); tr.optimizer.exp_avg_sq.copy_(opt_state[1]); tr.optimizer.step_count = opt_state[2]
                r.model._epoch += 1
                del junk
        return np.array(losses), r.model._flat.detach().clone()

    la, pa = run(False)
    lb, pb = run(True)
    assert np.all(np.isfinite(lb)) and np.allclose(la, lb, rtol=1e-6, atol=1e-7), (la, lb)
    assert float((pa - pb).abs().max()) <= 1e-6 * float(pa.abs().max())
