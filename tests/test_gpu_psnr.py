"""PSNR parity (BASELINE.json): the HIP renderer trained on the build-owned synthetic scene follows the PSNR curve of the
REFERENCE renderer trained (on CPU, tools/psnr_reference.py) with the same initial weights, batches and random draws."""
import os

import numpy as np
import pytest
import torch

import synth_scene
import weightgen
from gpu_util import renderer_for

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "psnr_reference.npz")


@pytest.mark.skipif(not os.path.exists(GOLD), reason="reference PSNR curve not generated")
def test_psnr_curve_matches_reference():
    from endosurf_amd.trainer import Trainer, cal_psnr
    g = np.load(GOLD)
    n_iter, n_rays = int(g["n_iter"]), int(g["n_rays"])
    ref_curve, ref_loss = g["curve"], g["loss"]
    r = renderer_for(int(g["weight_seed"]), "init", True)
    tr = Trainer(r, lr=5e-4, n_iter=n_iter, warm_up_end=max(n_iter // 10, 1), lr_alpha=0.05, fused=True)
    sched = synth_scene.schedule(int(g["sched_seed"]), n_iter, n_rays)
    ev = {k: torch.from_numpy(v).cuda() for k, v in synth_scene.eval_batch().items()}
    curve, losses = [], []
    for it in range(1, n_iter + 1):
        b = {k: torch.from_numpy(v).cuda() for k, v in sched[it - 1].items()}
        tr.update_learning_rate(it)
        loss, _, _ = tr.train_step(b, it, u_perturb=b["u_perturb"], u_neigh=b["u_neigh"])
        losses.append(loss)
        if it % 10 == 0 or it == 1:
            with torch.no_grad():
                e = r(ev["rays"], iter_step=it, perturb_overwrite=False)
            curve.append((it, float(cal_psnr(e["color_map"], ev["color"], ev["mask"]))))
    losses = torch.stack(losses).cpu().numpy()
    curve = np.array(curve)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    np.savez(os.path.join(out, "psnr_hip.npz"), curve=curve, loss=losses, ref_curve=ref_curve, ref_loss=ref_loss)
    assert np.array_equal(curve[:, 0], ref_curve[:, 0])
    d = curve[:, 1] - ref_curve[:, 1]
    # identical start (same weights), same trajectory early on, same quality at the end (training is chaotic in between)
    assert abs(d[0]) < 0.02, d[0]
    assert np.max(np.abs(d[: min(4, len(d))])) < 0.3, d[:4]
    # end quality: repeated runs of the SAME code end between ~27.9 and ~31.5 dB (fp32 atomics make the gradients
    # non-deterministic and 300 Adam steps amplify that), the reference's single CPU run ends at 30.3 dB
    end, ref_end = float(np.mean(curve[-3:, 1])), float(np.mean(ref_curve[-3:, 1]))
    assert abs(end - ref_end) < 3.0, (end, ref_end)
    assert curve[-1, 1] > curve[0, 1] + 3.0, "training must improve PSNR"
    assert abs(losses[0] - ref_loss[0]) < 2e-3 * max(1.0, abs(ref_loss[0]))
    assert abs(np.mean(losses[-20:]) - np.mean(ref_loss[-20:])) < 0.15 * abs(np.mean(ref_loss[-20:])) + 0.02
