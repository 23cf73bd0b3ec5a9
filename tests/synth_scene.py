"""Build-owned synthetic deforming scene for the PSNR-parity run (no datasets exist offline).

A textured sphere of radius r(t) = 0.55 + 0.08 sin(2 pi t), centred at (0, 0, 0.05 t), seen by the SURVEY 8d pinhole
camera.  Rays, colours and depths are analytic; depth is expressed in the reference's convention (multiples of
d / d.z along the ray, dataset.py:216-235 + endosurf.py:297).  Deterministic (numpy PCG64)."""
import numpy as np


def frame(t, n_rays, rng):
    u = rng.uniform(120, 520, size=n_rays)
    v = rng.uniform(56, 456, size=n_rays)
    d = np.stack([(u - 319.5) / 800.0, (v - 255.5) / 800.0, np.ones(n_rays)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.tile(np.array([[0.0, 0.0, -1.5]]), (n_rays, 1))
    c = np.array([0.0, 0.0, 0.05 * t])
    r = 0.55 + 0.08 * np.sin(2 * np.pi * t)
    oc = o - c
    b = (oc * d).sum(-1)
    disc = b * b - ((oc * oc).sum(-1) - r * r)
    hit = disc > 0
    s = -b - np.sqrt(np.maximum(disc, 0.0))          # distance along the unit direction
    p = o + s[:, None] * d
    n = (p - c) / r
    color = 0.5 + 0.5 * np.stack([np.sin(6 * n[:, 0] + 2 * t), np.cos(5 * n[:, 1]), np.sin(4 * n[:, 2] + n[:, 0])], -1)
    depth = s * d[:, 2]                               # z-depth: point = o + (d / d.z) * depth
    mask = hit.astype(np.float32)[:, None]
    rays = np.concatenate([o, d, np.zeros((n_rays, 2)), np.full((n_rays, 1), t)], -1)
    return dict(rays=rays.astype(np.float32), color=(color * mask).astype(np.float32), depth=(depth[:, None] * mask).astype(np.float32),
                mask=mask, color_mask=np.ones((n_rays, 1), np.float32))


def schedule(seed, n_iter, n_rays, n_frames=8):
    """Deterministic list of training batches + the uniform draws each iteration consumes."""
    rng = np.random.default_rng(seed)
    out = []
    for it in range(n_iter):
        t = float(rng.integers(0, n_frames)) / (n_frames - 1)
        b = frame(t, n_rays, rng)
        b["u_perturb"] = rng.uniform(size=(n_rays, 1)).astype(np.float32)
        b["u_neigh"] = rng.uniform(size=(n_rays, 3)).astype(np.float32)
        out.append(b)
    return out


def eval_batch(seed=999, n_rays=1024, t=0.5):
    return frame(t, n_rays, np.random.default_rng(seed))
