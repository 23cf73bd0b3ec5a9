#!/usr/bin/env python3
"""Golden vectors for the offline helpers of the renderer: renderonpts, renderondepth, extract_fields,
produced by running the REFERENCE implementation (build container only; see tools/make_golden.py for the import recipe).

    python tools/make_golden_offline.py      # writes tests/golden/offline_*.npz

Inputs come from the training-path fixtures (same seeds/weights); only inputs and reference outputs are stored."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import make_golden as MG
import weightgen  # noqa: E402  (tests/ is put on sys.path by make_golden)

RES = 20          # extract_fields resolution (one 20^3 block; the 128-blocking is exercised by the build's own test)


def run(E, cfg, state, dtype, c, tag, out):
    torch.set_default_dtype(dtype)
    r = MG.build_ref(E, cfg, state)
    if dtype == torch.float64:
        r = r.double()
        r.dtype = torch.float64
    cv = lambda a: torch.from_numpy(np.asarray(a)).to(dtype)
    npy = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    x, d, t = cv(c["pt/x"]), cv(c["pt/d"]), cv(c["pt/t"])
    with torch.no_grad():
        pass
    # renderonpts: per-point time [M,1] and the "one shared time" form ts.shape == [1]
    color, normal = r.renderonpts(x, d, t, cpu=True)
    out[f"onpts{tag}/color"] = npy(color); out[f"onpts{tag}/normal"] = npy(normal)
    t1 = cv(np.array([0.37]))
    color, normal = r.renderonpts(x.reshape(8, -1, 3), d.reshape(8, -1, 3), t1, cpu=False)
    out[f"onpts1{tag}/color"] = npy(color); out[f"onpts1{tag}/normal"] = npy(normal)
    # renderondepth on the ray-marching depths of the training fixture (contains 0 = no hit; add an inf and a negative)
    rays = cv(c["rays"])
    depth = cv(out["ondepth/depth_in"])
    col, grad, d_out = r.renderondepth(rays, depth)
    out[f"ondepth{tag}/color"] = npy(col); out[f"ondepth{tag}/gradients"] = npy(grad); out[f"ondepth{tag}/d_out"] = npy(d_out)
    none = torch.zeros_like(depth)
    col, grad, d_out = r.renderondepth(rays, none)
    out[f"ondepth_none{tag}/color"] = npy(col); out[f"ondepth_none{tag}/d_out"] = npy(d_out)
    # extract_fields (utils.py:139-157) through extract_observation_geometry's query function
    from src.renderer.utils import extract_fields, run_fn_split
    tq = cv(np.array([0.37]))        # 1-D: the "one time for all points" form (DeformNetwork.forward, endosurf.py:726-727)
    bmin, bmax = cv(np.array([-1.0, -0.9, -0.8])), cv(np.array([1.0, 0.9, 0.8]))
    q = lambda pts: run_fn_split(lambda p: r.model.get_sdf_from_observed_space(p, tq), pts, 3000, cpu=True)
    out[f"fields{tag}/u"] = extract_fields(bmin, bmax, RES, q, "cpu")
    torch.set_default_dtype(torch.float32)


def make(E, name, seed, use_deform):
    c = dict(np.load(os.path.join(MG.REPO, "tests", "golden", f"{name}.npz")))
    cfg = MG.load_cfg(use_deform)
    state = weightgen.make_state(seed, "trained", use_deform)
    out = {"meta/seed": np.array(seed), "meta/use_deform": np.array(use_deform), "meta/res": np.array(RES),
           "fields/bmin": np.array([-1.0, -0.9, -0.8], np.float32), "fields/bmax": np.array([1.0, 0.9, 0.8], np.float32),
           "fields/t": np.array(0.37, np.float32)}
    depth = c["march64/d_i"].astype(np.float32).copy()
    depth[1, 0] = np.inf
    depth[2, 0] = -0.25
    out["ondepth/depth_in"] = depth
    run(E, cfg, state, torch.float32, c, "", out)
    run(E, cfg, state, torch.float64, c, "64", out)
    for k in list(out):
        if out[k].dtype == np.float64 and out[k].size > 64:
            out[k] = out[k].astype(np.float32)
    path = os.path.join(MG.REPO, "tests", "golden", f"offline_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB; valid depths {int(((depth > 0) & np.isfinite(depth)).sum())}/{depth.shape[0]}")


if __name__ == "__main__":
    E = MG.import_reference()
    torch.set_num_threads(8)
    make(E, "trained_deform", 202, True)
    make(E, "trained_nodeform", 303, False)
