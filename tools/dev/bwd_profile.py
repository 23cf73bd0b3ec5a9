#!/usr/bin/env python3
"""Dev tool: phase cycle stamps of the SDF backward chain kernel (library built with -DES_PROFILE_BWD)."""
import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
import bench as B
from endosurf_amd import EndoSurfRenderer
from endosurf_amd.trainer import SyntheticScene, Trainer
dev = torch.device("cuda", 0)
cfg = B.CONFIGS[2]
r = EndoSurfRenderer(B.render_cfg(cfg), dict(B.NET_CFG), device=dev)
tr = Trainer(r); sc = SyntheticScene(dev, seed=1)
b = sc.batch(1024)
for i in range(3): tr.train_step(b, i + 1)
torch.cuda.synchronize()
buf = (C.c_longlong * 256)()
r.engine.lib.es_debug_b_profile.restype = C.c_int
r.engine.lib.es_debug_b_profile(buf, 256)
v = list(buf)
print("sdf_bwd tile total cycles", v[201] - v[200])
for l in range(1, 8):
    print(f"tangent layer {l}: gemm {v[4*l+1]-v[4*l]:6d} barrier {v[4*l+2]-v[4*l+1]:6d} epilogue {v[4*l+3]-v[4*l+2]:6d} barrier {(v[4*l+4] if l < 7 else 0)-v[4*l+3] if l<7 else -1:6d}")
for l in range(7, 0, -1):
    o = 100 + 4 * l
    print(f"reverse layer {l}: gemm {v[o+1]-v[o]:6d} barrier {v[o+2]-v[o+1]:6d} epilogue {v[o+3]-v[o+2]:6d}")
