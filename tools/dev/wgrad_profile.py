#!/usr/bin/env python3
"""Dev tool: phase cycle stamps of the fp32 weight-gradient task (library built with -DES_PROFILE_WGRAD)."""
import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
import bench as B
from endosurf_amd import EndoSurfRenderer
from endosurf_amd.trainer import SyntheticScene, Trainer
dev = torch.device("cuda", 0)
cfg = B.CONFIGS[2]
r = EndoSurfRenderer(B.render_cfg(cfg), dict(B.NET_CFG), device=dev)
tr = Trainer(r); b = SyntheticScene(dev, seed=1).batch(1024)
for i in range(3): tr.train_step(b, i + 1)
torch.cuda.synchronize()
buf = (C.c_longlong * 128)()
r.engine.lib.es_debug_w_profile.restype = C.c_int
r.engine.lib.es_debug_w_profile(buf, 128)
v = list(buf)
nst = v[20]
print(f"last launch (colour), task of block 0: {nst} stages of 16 rows; main loop {v[8]-v[0]} cycles = {(v[8]-v[0])/max(nst,1):.0f} per stage (2048 of own MFMA issue per wave, 4 waves per SIMD); epilogue {v[9]-v[8]}")
print(f"steady-state iteration: compute(0) {v[2]-v[1]}  sstore {v[3]-v[2]}  barrier {v[4]-v[3]}  compute(1) {v[5]-v[4]}  sstore {v[6]-v[5]}  barrier {v[7]-v[6]}")
