#!/usr/bin/env python3
"""A/B (DEAD_ENDS C6): the three weight-gradient launches of a step on three streams instead of one after the other."""
import sys, os, time, json, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from bench import Ctx, Workload
torch.cuda.set_device(0)
ctx = Ctx(torch.device("cuda", 0), 0, 1, False, False, "nccl", "fused")
wl = Workload(ctx, 2)
wl.eng.march_block = 0
def t(n=30):
    for i in range(5): wl.step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): wl.step(10 + i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
out = []
for rep in range(3):
    wl.eng._wgrad_streams_experiment = False; a = t()
    wl.eng._wgrad_streams_experiment = True; b = t()
    out.append((round(a, 3), round(b, 3)))
print("ms per step (one stream, three streams):", out)
