#!/usr/bin/env python3
"""Time the REFERENCE renderer on this container's CPU cores (build container only; SURVEY 8d / BASELINE.md 3).

    python tools/time_reference_cpu.py [--threads 8] [--warmup 3] [--iters 5]   -> profiles/reference_cpu.json

What is timed, on the synthetic inputs of SURVEY 8d (tests/weightgen.py rays/targets, reference-initialised weights under
torch.manual_seed(0), perturb=True, iter_step=1):
  (A) forward only: ``EndoSurfRenderer(rays, iter_step=1)`` under no_grad
  (B) full training step (trainer_endosurf.py:94-162 restated: render + errorondepth + surface_neighbour_error + loss +
      backward + Adam; no logging)
at 256 rays (BASELINE config 1) and 1024 rays (config 2), plus config 4 (use_deform False) at 1024 rays.  ``warmup``
untimed iterations, then ``iters`` timed ones; the median is reported.  The reference is imported unmodified with the two
stub modules of tools/make_golden.py; nothing of it is copied and it never travels to the GPU box.
"""
import argparse
import json
import os
import platform
import sys
import time

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np
import torch

import make_golden as MG
import weightgen


def build(E, use_deform):
    import warnings
    cfg = MG.load_cfg(use_deform)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = E.EndoSurfRenderer(cfg["render"], cfg["net"], device="cpu")
    return r


def make_batch(n):
    rays = torch.from_numpy(weightgen.make_rays(1, n))
    tg = {k: torch.from_numpy(v) for k, v in weightgen.make_targets(2, n).items()}
    return dict(rays=rays, **tg)


def train_step(r, opt, b, it):
    opt.zero_grad()
    ret = r(b["rays"], iter_step=it)
    color_loss = ((ret["color_map"] - b["color"]) * b["color_mask"]).abs().sum() / (b["color_mask"].sum() + 1e-10)
    sdf_loss, angle_loss, vd = r.errorondepth(b["rays"], d_gt=b["depth"], mask=b["mask"], iter_step=it)
    depth_loss = ((ret["depth_map"] - b["depth"]) * vd * b["mask"]).abs().sum() / ((vd * b["mask"]).sum() + 1e-10)
    sn = r.surface_neighbour_error(rays=b["rays"], mask=b["mask"], iter_step=it, neighbour_rad=0.1)
    loss = color_loss + depth_loss + sdf_loss + 0.1 * angle_loss + 0.1 * ret["gradient_o_error"] + 0.1 * sn
    loss.backward()
    opt.step()
    return float(loss)


def timed(fn, warmup, iters):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "reference_cpu.json"))
    args = ap.parse_args()
    E = MG.import_reference()
    torch.set_num_threads(args.threads)
    rows = []
    for label, use_deform, n in (("cfg1 base_pull 256 rays", True, 256), ("cfg2 base_pull 1024 rays", True, 1024),
                                 ("cfg4 base_d1k1 (use_deform False) 1024 rays", False, 1024)):
        r = build(E, use_deform)
        b = make_batch(n)

        def fwd():
            with torch.no_grad():
                r(b["rays"], iter_step=1)
        ts = timed(fwd, args.warmup, args.iters)
        rows.append(dict(config=label, what="forward (renderer(rays, iter_step=1), no_grad)", n_rays=n, seconds=ts,
                         median_s=float(np.median(ts)), rays_per_s=n / float(np.median(ts))))
        print(rows[-1]["config"], rows[-1]["what"], f"{rows[-1]['median_s']:.3f} s  {rows[-1]['rays_per_s']:.1f} rays/s", flush=True)
        opt = torch.optim.Adam([p for p in r.parameters()], lr=5e-4)
        it = [0]

        def step():
            it[0] += 1
            train_step(r, opt, b, it[0])
        ts = timed(step, args.warmup, args.iters)
        rows.append(dict(config=label, what="full training step (render + errorondepth + surface_neighbour_error + loss + backward + Adam)",
                         n_rays=n, seconds=ts, median_s=float(np.median(ts)), rays_per_s=n / float(np.median(ts))))
        print(rows[-1]["config"], rows[-1]["what"], f"{rows[-1]['median_s']:.3f} s  {rows[-1]['rays_per_s']:.1f} rays/s", flush=True)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    out = dict(what="reference EndoSurfRenderer (Ruyi-Zha/endosurf, imported unmodified) on the build container's CPU",
               torch=torch.__version__, threads=torch.get_num_threads(), vcpus=os.cpu_count(), cpu=cpu, machine=platform.machine(),
               dtype="float32", warmup=args.warmup, iters=args.iters, samples_per_ray="32 coarse + 32 importance", rows=rows)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
