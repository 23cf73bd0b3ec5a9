#!/usr/bin/env python3
"""Dev tool: phase cycle stamps of the fp32 64-point SDF query kernel (library built with -DES_PROFILE_QUERY)."""
import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
from gpu_util import renderer_for
r = renderer_for(24, "trained", True)
x = torch.rand(163840, 3, device="cuda") - 0.5; t = torch.rand(163840, device="cuda")
for _ in range(3): r.sdf_observed(x, t)
torch.cuda.synchronize()
buf = (C.c_longlong * 64)()
r.engine.lib.es_debug_q_profile.restype = C.c_int
r.engine.lib.es_debug_q_profile(buf, 64)
v = list(buf)
names = ["load + deform encode", "deform layer 0", "deform layers 1-7", "deform tail (3 outputs) + x_c", "sdf encode + layer 0", "sdf layers 1-7", "sdf tail", ]
tot = v[7] - v[0]
print("tile total cycles", tot)
for i, n in enumerate(names):
    print(f"{n:32s} {v[i+1]-v[i]:8d}  {100*(v[i+1]-v[i])/tot:5.1f} %")
print("deform layer gemm-only cycles:", [v[20+l]-v[10+l] for l in range(1, 8)])
