#!/usr/bin/env python3
"""Memory after many iterations of the reference trainer's own loop (bench.ReferenceLoop: tail + deferred errorondepth): must be flat."""
import sys, os, gc, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from bench import Ctx, ReferenceLoop
torch.cuda.set_device(0)
ctx = Ctx(torch.device("cuda", 0), 0, 1, False, False, "nccl", "plain")
for logging in (False, True):
    loop = ReferenceLoop(ctx, logging=logging)
    marks = []
    for i in range(601):
        loop.train_step(i)
        if i in (50, 200, 400, 600):
            torch.cuda.synchronize()
            marks.append((i, torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, len(gc.get_objects())))
    print("logging" if logging else "plain", marks)
    assert marks[-1][1] <= marks[0][1] + 64, marks
    loop = None; gc.collect(); torch.cuda.empty_cache()
