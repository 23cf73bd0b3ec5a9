#!/usr/bin/env python3
"""Golden vectors for the dataset-side callers of the hot path: pinhole ray generation, the mask-guided
inverse-CDF pixel sampler and the training-batch gather of the REFERENCE Dataset class (src/dataset/dataset.py), run in the
build container on a small synthetic frame set.  The class is imported with stub modules for its absent visualisation
dependencies (imageio, cv2, open3d, src.trainer.utils) and instantiated without __init__ (which only reads files).

    python tools/make_golden_data.py        # writes tests/golden/data_small.npz"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_ROOT = os.environ.get("ENDOSURF_REFERENCE", "/root/reference")

import numpy as np
import torch


def import_dataset():
    for name in ("imageio", "imageio.v2", "cv2", "open3d"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["imageio"].v2 = sys.modules["imageio.v2"]
    tr = types.ModuleType("src.trainer")
    tr.__path__ = []
    stub = types.ModuleType("src.trainer.utils")
    stub.gen_pcd = stub.to8b = None
    sys.modules["src.trainer"] = tr
    sys.modules["src.trainer.utils"] = stub
    os.chdir(REF_ROOT)
    sys.path.insert(0, REF_ROOT)
    from src.dataset.dataset import Dataset
    return Dataset


def rigid(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = q
    m[:3, 3] = rng.normal(size=3) * 0.3
    return m


def main():
    Dataset = import_dataset()
    rng = np.random.default_rng(77)
    n, h, w, B = 3, 12, 16, 40
    K = np.tile(np.eye(4, dtype=np.float32)[None], (n, 1, 1))
    for i in range(n):
        K[i, 0, 0], K[i, 1, 1], K[i, 0, 2], K[i, 1, 2] = 20.0 + i, 21.0 + i, 7.5 + 0.1 * i, 5.5 - 0.1 * i
    poses = np.stack([rigid(rng) for _ in range(n)])
    colors = rng.uniform(size=(n, h, w, 3)).astype(np.float32)
    depths = (1.0 + rng.uniform(size=(n, h, w, 1))).astype(np.float32)
    depth_masks = (rng.uniform(size=(n, h, w, 1)) > 0.15).astype(np.float32)
    color_masks = (rng.uniform(size=(n, h, w, 1)) > 0.25).astype(np.float32)
    bounds = np.array([[0.1, 2.0]] * n, np.float32)
    u = rng.uniform(size=(1, B)).astype(np.float32)
    out = dict(K=K, poses=poses, colors=colors, depths=depths, depth_masks=depth_masks, color_masks=color_masks, bounds=bounds, u=u,
               meta=np.array([n, h, w, B]))

    ds = Dataset.__new__(Dataset)
    ds.device = "cpu"
    ds.h, ds.w, ds.n_frames = h, w, n
    T = torch.from_numpy
    rays = ds.get_rays(T(K), T(poses), w, h)
    out["rays6"] = rays.numpy()
    ds.colors, ds.depths, ds.depth_masks, ds.color_masks = T(colors), T(depths), T(depth_masks), T(color_masks)
    ds.masks = ds.depth_masks * ds.color_masks
    bds = T(bounds)[:, None, None, :].expand(n, h, w, 2)
    ts = torch.linspace(0., 1., n)[:, None, None, None].expand(n, h, w, 1)
    ds.rays = torch.cat([rays, bds, ts], -1)
    out["rays9"] = ds.rays.numpy()
    ds.list_train = [0, 2]
    ds.ray_importance_maps = Dataset._ray_sampling_importance_from_masks(ds.masks)
    out["importance"] = ds.ray_importance_maps.numpy()
    wts = T(rng.uniform(size=(1, 57)).astype(np.float32))
    out["is_weights"] = wts.numpy()
    out["is_det"] = Dataset._importance_sampling_coords(wts, 9, det=True, device="cpu").numpy()
    real_rand = torch.rand
    try:
        torch.rand = lambda *a, **k: T(u).clone()
        out["is_u"] = Dataset._importance_sampling_coords(wts, B, det=False, device="cpu").numpy()
        for fid in ds.list_train:
            b = ds.get_train_batch_data_by_index(fid, ray_batch=B, mask_guided_ray_sampling=True)
            for k, v in b.items():
                out[f"batch{fid}/{k}"] = v.numpy()
    finally:
        torch.rand = real_rand
    path = os.path.join(REPO, "tests", "golden", "data_small.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
