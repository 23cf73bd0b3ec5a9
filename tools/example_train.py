#!/usr/bin/env python3
"""End-to-end example on the build-owned synthetic scene (GPU box): train -> reference-format checkpoint -> resume -> full-frame
render -> mesh.  Everything a user of the reference's trainer touches, through the drop-in's public surface.

    python tools/example_train.py [--iters 300] [--rays 1024] [--out gpurun_out/example]
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch

import bench as B
import synth_scene
from endosurf_amd import EndoSurfRenderer
from endosurf_amd.trainer import Trainer, cal_psnr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "example"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = B.CONFIGS[2]
    renderer = EndoSurfRenderer(B.render_cfg(cfg), dict(B.NET_CFG), device=dev)
    trainer = Trainer(renderer, n_iter=args.iters, warm_up_end=max(args.iters // 10, 1))
    sched = synth_scene.schedule(5, args.iters, args.rays)
    ev = {k: torch.from_numpy(v).to(dev) for k, v in synth_scene.eval_batch().items()}
    half = args.iters // 2
    t0 = time.perf_counter()
    for it in range(1, half + 1):
        b = {k: torch.from_numpy(v).to(dev) for k, v in sched[it - 1].items()}
        trainer.update_learning_rate(it)
        loss, terms, _ = trainer.train_step(b, it)
    # checkpoint in the reference's ckpt.tar format, resume in a NEW renderer / trainer
    path = os.path.join(args.out, "ckpt.tar")
    torch.save(trainer.save_checkpoint(half), path)
    renderer2 = EndoSurfRenderer(B.render_cfg(cfg), dict(B.NET_CFG), device=dev)
    trainer2 = Trainer(renderer2, n_iter=args.iters, warm_up_end=max(args.iters // 10, 1))
    start = trainer2.load_checkpoint(torch.load(path, weights_only=False))
    assert start == half + 1
    for it in range(start, args.iters + 1):
        b = {k: torch.from_numpy(v).to(dev) for k, v in sched[it - 1].items()}
        trainer2.update_learning_rate(it)
        loss, terms, _ = trainer2.train_step(b, it)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    with torch.no_grad():
        e = renderer2(ev["rays"], iter_step=args.iters, perturb_overwrite=False)
    psnr = float(cal_psnr(e["color_map"], ev["color"], ev["mask"]))
    print(f"{args.iters} iterations x {args.rays} rays in {dt:.1f} s ({args.iters * args.rays / dt:.0f} rays/s incl. host data prep), "
          f"final loss {float(loss):.4f}, eval PSNR {psnr:.2f} dB")
    # a full 640x512 frame through the hipGraph-replayed chunk renderer
    from endosurf_amd.trainer import SyntheticScene
    frame_rays = SyntheticScene(dev).frame(t=0.5)
    t1 = time.perf_counter()
    img = renderer2.render_frames(frame_rays, iter_step=args.iters, ray_chunk=2048, perturb_overwrite=False)
    torch.cuda.synchronize()
    print(f"frame 640x512: {time.perf_counter() - t1:.2f} s (first call captures the graph), colour {tuple(img['color'].shape)}, "
          f"depth range {float(img['depth'].min()):.2f}..{float(img['depth'].max()):.2f}")
    np.save(os.path.join(args.out, "frame_color.npy"), img["color"].reshape(512, 640, 3).cpu().numpy().astype(np.float16))
    # observed-space mesh at t = 0.5 (field sampled on the GPU; PyMCubes if installed, else marching tetrahedra)
    v, f = renderer2.extract_observation_geometry(torch.tensor([0.5]), [-1, -1, -1], [1, 1, 1], resolution=96)
    print(f"mesh: {len(v)} vertices, {len(f)} triangles")
    assert np.isfinite(psnr) and len(v) > 0


if __name__ == "__main__":
    main()
