"""Dataset-side helpers (endosurf_amd/data.py) against vectors captured from the reference Dataset class
(tests/golden/data_small.npz, tools/make_golden_data.py).  CPU only."""
import os

import numpy as np
import torch

from endosurf_amd import data as D

G = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data_small.npz")))
T = torch.from_numpy


def test_get_rays_and_ray_tensor_contract():
    n, h, w, _ = (int(v) for v in G["meta"])
    rays6 = D.get_rays(T(G["K"]), T(G["poses"]), w, h)
    assert tuple(rays6.shape) == (n, h, w, 6)
    assert np.abs(rays6.numpy() - G["rays6"]).max() < 2e-6
    assert np.abs(np.linalg.norm(rays6[..., 3:].numpy(), axis=-1) - 1).max() < 1e-6
    rays9 = D.assemble_rays(rays6, T(G["bounds"]))
    assert np.abs(rays9.numpy() - G["rays9"]).max() < 2e-6
    assert np.array_equal(rays9[..., 8].numpy()[:, 0, 0], np.linspace(0, 1, n, dtype=np.float32))


def test_importance_maps_and_sampler():
    masks = T(G["depth_masks"]) * T(G["color_masks"])
    imp = D.ray_sampling_importance_from_masks(masks)
    assert np.abs(imp.numpy() - G["importance"]).max() < 1e-6
    wts = T(G["is_weights"])
    assert np.array_equal(D.importance_sampling_coords(wts, 9, det=True).numpy(), G["is_det"])
    assert np.array_equal(D.importance_sampling_coords(wts, G["u"].shape[1], u=T(G["u"])).numpy(), G["is_u"])


def test_train_batch_matches_reference_selection():
    n, h, w, B = (int(v) for v in G["meta"])
    fs = D.FrameSet(G["colors"], G["depths"], G["K"], G["poses"], G["bounds"], color_masks=G["color_masks"],
                    depth_masks=G["depth_masks"], list_train=[0, 2], device="cpu")
    assert np.abs(fs.rays.numpy() - G["rays9"]).max() < 2e-6
    for fid in (0, 2):
        b = fs.get_train_batch_data_by_index(fid, ray_batch=B, u=T(G["u"]))
        # the static-shape sampler (zero weight outside the colour mask) must select the same pixels as the reference's
        # compact-then-sample, except where a uniform draw lands within float rounding of a CDF step
        same = np.all(np.abs(b["rays"].numpy() - G[f"batch{fid}/rays"]) < 2e-6, axis=-1)
        assert same.mean() >= 0.95, same.mean()
        for k in ("color", "depth", "mask", "color_mask", "depth_mask"):
            assert np.array_equal(b[k].numpy()[same], G[f"batch{fid}/{k}"][same]), k
        assert (b["color_mask"] == 1).all()                       # never draws a pixel outside the colour mask
    b = fs.get_train_batch_data_by_index(None, ray_batch=64, mask_guided_ray_sampling=False)
    assert b["rays"].shape == (64, 9) and (b["color_mask"] == 1).all()
    fr = fs.get_frame_data_by_index(1)
    assert fr["rays"].shape == (h, w, 9)


def test_metrics():
    rng = np.random.default_rng(0)
    a, b = rng.uniform(size=(4, 5, 3)), rng.uniform(size=(4, 5, 3))
    m = (rng.uniform(size=(4, 5)) > 0.3).astype(np.float64)
    mse = ((a - b) ** 2 * m[..., None]).sum() / (m.sum() * 3)
    assert abs(D.cal_psnr(a, b, m) - (-10 * np.log10(mse))) < 1e-9
    assert abs(D.cal_psnr(T(a), T(b), T(m)) - D.cal_psnr(a, b, m)) < 1e-9
    assert abs(D.cal_rmse(a[..., :1], b[..., :1], m) - np.sqrt(((a[..., :1] - b[..., :1]) ** 2 * m[..., None]).sum() / m.sum())) < 1e-9
