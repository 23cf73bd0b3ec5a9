"""Dataset-side callers of the hot path: what the reference's ``Dataset`` (src/dataset/dataset.py) does
between decoded images and the renderer — pinhole ray generation, the [n,h,w,9] ray tensor contract, the mask-guided
inverse-CDF pixel sampler and the per-iteration batch gather — as device-side torch code without host synchronisation.
File decoding (imageio / cv2 / pickle info files) is out of scope: a ``FrameSet`` is built from arrays already in memory."""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch


def get_rays(intrinsics: torch.Tensor, poses: torch.Tensor, w: int, h: int) -> torch.Tensor:
    """Dataset.get_rays (dataset.py:216-235): per frame d = normalize(K^-1 [x, y, 1]) rotated to world, o = pose translation.
    intrinsics, poses: [n,4,4] -> rays [n,h,w,6] (origin, unit direction)."""
    dev, dt = intrinsics.device, intrinsics.dtype
    k_inv = torch.inverse(intrinsics)[:, :3, :3]
    ys, xs = torch.meshgrid(torch.linspace(0, h - 1, h, device=dev, dtype=dt), torch.linspace(0, w - 1, w, device=dev, dtype=dt),
                            indexing="ij")
    p = torch.stack([xs, ys, torch.ones_like(xs)], -1)                               # [h,w,3]
    d = torch.einsum("nij,hwj->nhwi", k_inv, p)
    d = d / torch.linalg.norm(d, ord=2, dim=-1, keepdim=True)
    d = torch.einsum("nij,nhwj->nhwi", poses[:, :3, :3], d)
    o = poses[:, None, None, :3, 3].expand_as(d)
    return torch.cat([o, d], -1)


def assemble_rays(rays6: torch.Tensor, bounds: torch.Tensor, normalize_time: bool = True) -> torch.Tensor:
    """[n,h,w,9] = rays6 | per-frame (near, far) bounds | per-frame time (dataset.py:86-96); time = linspace(0,1,n) or the index."""
    n, h, w, _ = rays6.shape
    ts = torch.linspace(0.0, 1.0, n, device=rays6.device) if normalize_time else torch.arange(n, device=rays6.device, dtype=rays6.dtype)
    return torch.cat([rays6, bounds[:, None, None, :].expand(n, h, w, 2).to(rays6.dtype),
                      ts[:, None, None, None].expand(n, h, w, 1).to(rays6.dtype)], -1)


def ray_sampling_importance_from_masks(masks: torch.Tensor) -> torch.Tensor:
    """Dataset._ray_sampling_importance_from_masks (dataset.py:262-267).  masks [n,h,w,1] -> importance [n,h,w,1]:
    a visible pixel weighs 1 + (how often that pixel is masked out over the sequence, L2-normalised over the image)."""
    hidden = masks.shape[0] - masks.sum(dim=0)                 # per pixel: number of frames that mask it out
    return masks * (1.0 + hidden / torch.linalg.vector_norm(hidden))


def sampling_cdf(weights: torch.Tensor, floor: float = 1e-5) -> torch.Tensor:
    """Normalised running sum of ``weights + floor`` along the last axis: the table the inverse-CDF draw searches."""
    w = weights + floor
    return torch.cumsum(w / w.sum(dim=-1, keepdim=True), dim=-1)


def importance_sampling_coords(weights: torch.Tensor, n_samples: int, det: bool = False, u: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Dataset._importance_sampling_coords (dataset.py:237-260): ``n_samples`` indices per row of ``weights`` drawn by inverting
    the CDF of (weights + 1e-5) with searchsorted(right=True).  ``det``: evenly spaced quantiles; ``u``: caller-supplied uniform
    draws (reproducible tests); otherwise torch.rand on the weights' device."""
    cdf = sampling_cdf(weights)
    rows = tuple(cdf.shape[:-1])
    if det:
        u = torch.linspace(0.0, 1.0, n_samples, device=cdf.device).expand(*rows, n_samples)
    elif u is None:
        u = torch.rand(*rows, n_samples, device=cdf.device)
    return torch.searchsorted(cdf, u.contiguous(), right=True)


class FrameSet:
    """In-memory stand-in for the reference Dataset's tensors: colors [n,h,w,3], depths [n,h,w,1] (already scaled), masks.
    ``get_train_batch_data_by_index`` mirrors dataset.py:117-161 (one frame per iteration, pixels outside the colour mask are
    never drawn, mask-guided importance sampling by default) and stays on the device."""

    def __init__(self, colors, depths, intrinsics, poses, bounds, color_masks=None, depth_masks=None, near=None, far=None,
                 normalize_time: bool = True, list_train: Optional[Sequence[int]] = None, device="cuda"):
        dev = torch.device(device)
        T = lambda a: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a, dtype=torch.float32).to(dev)
        self.device = dev
        self.colors, self.depths = T(colors), T(depths)
        self.n_frames, self.h, self.w = self.colors.shape[:3]
        if depth_masks is None:       # dataset.py:75-77: depths inside the [3 %, 99.5 %] percentile range are trusted
            d = self.depths.detach().cpu().numpy()
            self.near = float(np.percentile(d, 3.0)) if near is None else near
            self.far = float(np.percentile(d, 99.5)) if far is None else far
            depth_masks = ((self.depths > self.near) & (self.depths < self.far)).to(torch.float32)
        self.depth_masks = T(depth_masks)
        self.color_masks = T(color_masks) if color_masks is not None else torch.ones_like(self.depth_masks)
        self.masks = self.depth_masks * self.color_masks
        self.intrinsics, self.poses = T(intrinsics), T(poses)
        self.rays = assemble_rays(get_rays(self.intrinsics, self.poses, self.w, self.h), T(bounds), normalize_time)
        self.ray_importance_maps = ray_sampling_importance_from_masks(self.masks)
        self.list_train = list(range(self.n_frames)) if list_train is None else list(list_train)
        self._host_rng = np.random.default_rng(0)
        # per-frame sampling tables (the CDF over the colour-masked pixels, the index of the last kept pixel, the kept-pixel lists of
        # the uniform branch) are built lazily, one frame at a time, the first time a frame is drawn: no [n_frames, H*W] temporaries
        # at construction and no host round trip per batch afterwards
        self._tables = {}
        self._kept = {}

    def _frame_tables(self, i: int):
        """(cdf [H*W] fp32, last kept pixel [] int64, colour-mask [H*W] bool) of frame ``i``; raises if its colour mask is empty
        (the reference's compaction would fail there as well, dataset.py:131-137)."""
        t = self._tables.get(i)
        if t is None:
            cm = self.color_masks[i, ..., 0].reshape(-1) == 1.0
            imp = self.ray_importance_maps[i, ..., 0].reshape(-1)
            # sampling over the colour-masked pixels only == sampling over all pixels with the others' weight removed; a zero weight
            # (instead of the reference's compaction) keeps shapes static.  The 1e-5 floor is applied to kept pixels only
            wts = torch.where(cm, imp + 1e-5, torch.zeros_like(imp))
            total = wts.sum()
            if not bool(total > 0):
                raise ValueError(f"frame {i}: the colour mask is empty, there is no pixel to sample")
            idx = torch.arange(cm.shape[0], device=self.device, dtype=torch.int32)
            last = torch.where(cm, idx, torch.zeros_like(idx)).amax().to(torch.int64)
            t = self._tables[i] = (torch.cumsum(wts / total, -1), last, cm)
        return t

    def get_train_batch_data_by_index(self, id_train=None, ray_batch=1024, mask_guided_ray_sampling=True, u=None) -> Dict[str, torch.Tensor]:
        if id_train is None:
            id_train = int(self._host_rng.choice(self.list_train))
        else:
            assert id_train in self.list_train, f"ID {id_train} is not in training list!"
        if mask_guided_ray_sampling:
            if u is None:
                u = torch.rand(ray_batch, device=self.device)
            cdf, last_kept, _ = self._frame_tables(id_train)
            sel = torch.searchsorted(cdf, u.reshape(-1).to(self.device).contiguous(), right=True)
            # clamp like the reference (max with 0, min with last kept pixel): u -> 1 rounds to the last colour-masked pixel
            sel = torch.minimum(sel, last_kept)
        else:
            kept = self._kept.get(id_train)
            if kept is None:
                kept = self._kept[id_train] = torch.nonzero(self._frame_tables(id_train)[2]).reshape(-1)      # once per frame
            sel = kept[torch.randperm(kept.numel(), device=self.device)[:ray_batch]]
        pick = lambda a: a[id_train].reshape(self.h * self.w, -1)[sel]
        return {"color": pick(self.colors), "rays": pick(self.rays), "depth": pick(self.depths), "mask": pick(self.masks),
                "color_mask": pick(self.color_masks), "depth_mask": pick(self.depth_masks)}

    def get_frame_data_by_index(self, idx):
        """dataset.py:163-181: whole frames (for evaluation / render_frames)."""
        return {"color": self.colors[idx], "rays": self.rays[idx], "depth": self.depths[idx], "mask": self.masks[idx],
                "color_mask": self.color_masks[idx], "depth_mask": self.depth_masks[idx]}


def cal_psnr(a, b, mask) -> float:
    """src/trainer/utils.py:340-354."""
    a, b, mask = (x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x) for x in (a, b, mask))
    if mask.ndim == a.ndim - 1:
        mask = mask[..., None]
    return float(20.0 * np.log10(1.0 / (((a - b) ** 2 * mask).sum() / ((np.sum(mask) + 1e-10) * 3.0)) ** 0.5))


def cal_rmse(a, b, mask) -> float:
    """src/trainer/utils.py:357-370."""
    a, b, mask = (x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x) for x in (a, b, mask))
    if mask.ndim == a.ndim - 1:
        mask = mask[..., None]
    return float((((a - b) ** 2 * mask).sum() / (np.sum(mask) + 1e-10)) ** 0.5)
