#!/usr/bin/env python3
"""Dev tool (library built with -DES_DEV_SWITCHES): the SDF query kernel under the variant ES_QT selects -- outputs of a few batch
shapes dumped for a bit-wise comparison between variants, and the launch time of the two sizes a training step runs.

    ES_QT=0 python tools/dev/qt_ab.py a; ES_QT=1 python tools/dev/qt_ab.py b; python tools/dev/qt_ab.py cmp a b"""
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)

if sys.argv[1] == "cmp":
    a, b = (np.load(os.path.join(OUT, "qt_%s.npz" % t)) for t in sys.argv[2:4])
    ok = True
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        ok &= same
        print(k, a[k].shape, "bit-identical" if same else "DIFFERENT: max |d| = %.3e, %d of %d" % (np.abs(a[k] - b[k]).max(), (a[k] != b[k]).sum(), a[k].size))
    sys.exit(0 if ok else 1)

import torch
from gpu_util import renderer_for
from endosurf_amd.trainer import SyntheticScene

tag = sys.argv[1]
res, times = {}, {}
for use_deform in (True, False):
    r = renderer_for(24, "trained", use_deform)
    eng = r.engine
    weff, packed = r._weights()
    weff = weff.detach()
    rays = SyntheticScene("cuda", seed=5).batch(1024)["rays"].float().contiguous()
    z128 = eng.empty(1024, 128)
    eng.ray_setup(rays, None, 128, 0.0, 1, z128)
    z32 = z128[:, ::4].contiguous()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.rand(5000, 3, device="cuda", generator=g) - 0.5
    t = torch.rand(5000, device="cuda", generator=g)
    cases = {"march_131072_t64": (lambda: eng.points(rays=rays, z=z128, n_per_ray=128, ldz=128), 64),
             "coarse_32768_t32": (lambda: eng.points(rays=rays, z=z32, n_per_ray=32, ldz=32), 32),
             "coarse_32768_t64": (lambda: eng.points(rays=rays, z=z32, n_per_ray=32, ldz=32), 64),
             "ragged_5000_t64": (lambda: eng.points(x=x, t=t), 64),
             "ragged_4999_t32": (lambda: eng.points(x=x[:4999].contiguous(), t=t[:4999].contiguous()), 32),
             "one_point_t64": (lambda: eng.points(x=x[:1].contiguous(), t=t[:1].contiguous()), 64)}
    for name, (mk, tile) in cases.items():
        out = eng.query_sdf(mk(), weff, packed, use_deform, tile_points=tile)
        torch.cuda.synchronize()
        res["%s_deform%d" % (name, use_deform)] = out.cpu().numpy()
        if name.startswith(("march", "coarse")):
            for _ in range(3):
                eng.query_sdf(mk(), weff, packed, use_deform, tile_points=tile)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            a.record()
            for _ in range(n):
                eng.query_sdf(mk(), weff, packed, use_deform, tile_points=tile)
            b.record(); torch.cuda.synchronize()
            times["%s_deform%d" % (name, use_deform)] = a.elapsed_time(b) / n
    # the full marching path (early-exit tiles, [ray][ld] outputs) and a forward render
    d = r.ray_marching(rays)
    res["ray_marching_deform%d" % use_deform] = d.cpu().numpy()
    with torch.no_grad():
        o = r.render_rays(rays, iter_step=100, perturb_overwrite=False)
    res["render_depth_deform%d" % use_deform] = o["depth_map"].cpu().numpy()
np.savez(os.path.join(OUT, "qt_%s.npz" % tag), **res)
print(tag, "ES_QT=%s" % os.environ.get("ES_QT"), json.dumps({k: round(v, 4) for k, v in times.items()}))
