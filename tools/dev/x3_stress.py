"""Dev tool: race screen of the split-precision inference chain -- large batches, several repetitions, compared element by element with
the fp32 kernels; a staging race would show up as isolated large errors that come and go between repetitions."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch
from test_gpu_point import _setup
from endosurf_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for seed, use_deform in ((31, True), (32, True), (33, False)):
    eng, flat, weff, packed, net = _setup(seed, "trained", use_deform)
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.uniform(-0.8, 0.8, size=(M, 3)).astype(np.float32)).cuda()
    d = rng.normal(size=(M, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = torch.from_numpy(d.astype(np.float32)).cuda()
    t = torch.from_numpy(rng.uniform(size=(M,)).astype(np.float32)).cuda()
    flags = (_lib.PF_DEFORM if use_deform else 0) | _lib.PF_COLOR
    eng.split_precision = False
    ref = eng.point_forward(eng.points(x=x, t=t, dirs=d), weff, packed, flags)
    torch.cuda.synchronize()
    R = {k: ref.view(k).clone() for k in ("xc", "sdf", "gc", "go", "rgb", "feat")}
    eng.split_precision, eng.x3_infer_min = True, 1
    prev = None
    for rep in range(6):
        ctx = eng.point_forward(eng.points(x=x, t=t, dirs=d), weff, packed, flags)
        torch.cuda.synchronize()
        cur = {k: ctx.view(k).clone() for k in R}
        line = []
        for k in R:
            e = (cur[k] - R[k]).abs()
            line.append("%s max %.2e q9999 %.2e" % (k, float(e.max()), float(torch.quantile(e.flatten()[:4000000], 0.9999))))
        same = prev is None or all(torch.equal(cur[k], prev[k]) for k in R)
        print("seed", seed, "deform", use_deform, "rep", rep, "bit-identical to previous rep:", same, "|", " | ".join(line))
        prev = cur
