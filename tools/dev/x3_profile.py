import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
from gpu_util import renderer_for
r = renderer_for(24, "trained", True); r.engine.split_precision = True
x = torch.rand(163840, 3, device="cuda") - 0.5; t = torch.rand(163840, device="cuda")
for _ in range(3): r.sdf_observed(x, t)
torch.cuda.synchronize()
buf = (C.c_longlong * 512)()
r.engine.lib.es_debug_x3_profile.restype = C.c_int
print("rc", r.engine.lib.es_debug_x3_profile(buf, 512))
v = list(buf)
print("total cycles block0", v[140] - v[0])
for l in range(1, 8):
    b = 100 + 4 * l
    print(f"sdf layer {l}: gemm {v[b+1]-v[b]}  barrier {v[b+2]-v[b+1]}  epilogue {v[b+3]-v[b+2]}  next-barrier {(v[b+4] if l<7 else v[140])-v[b+3]}")
