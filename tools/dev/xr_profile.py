import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
from gpu_util import renderer_for
r = renderer_for(24, "trained", True); r.engine.split_precision = True
M = 131072
x = torch.rand(M, 3, device="cuda") - 0.5; t = torch.rand(M, device="cuda")
for _ in range(3): r.sdf_observed(x, t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
lib = r.engine.lib
lib.es_debug_xr_reset()
e0.record()
for _ in range(10): r.sdf_observed(x, t)
e1.record(); torch.cuda.synchronize()
print("ms per 131072-point query", e0.elapsed_time(e1) / 10)
buf = (C.c_longlong * 512)()
lib.es_debug_xr_profile(buf, 512)
v = list(buf)
print("block 0 total", v[19] - v[0], "prologue+deform", v[1] - v[0], "sdf layer0", v[11] - v[1])
print("sdf layers 1..7:", [v[11 + l] - v[10 + l] for l in range(1, 8)], "tail", v[19] - v[18])
if v[301]:
    print("acquire wait cycles per k-step", v[300] / v[301], "k-steps", v[301] / 10)
