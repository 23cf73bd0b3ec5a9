#!/usr/bin/env python3
"""Dev tool (GPU box): the whole front end of a training step (128-proposal marching query, secant chain, sampling chain) under two
schedules: S1 = as shipped (sampling chain starts on the side stream AFTER the marching query), S2 = the sampling chain starts together with
the marching query.  Prints ms per front end."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
from endosurf_amd import EndoSurfRenderer
from endosurf_amd.trainer import SyntheticScene
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg = B.CONFIGS[2]
r = EndoSurfRenderer(B.render_cfg(cfg), dict(B.NET_CFG), device=dev)
r.engine.march_block = 0
sc = SyntheticScene(dev, seed=1234)
rays = r._rays32(sc.batch(cfg["rays"])["rays"])
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)
r._weights()

def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(n):
        fn()
    b.record(main)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n

def s1():
    ms = r._march_begin(rays)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        r.sample_z(rays, 1)
    r._march_refine(ms)
    main.wait_stream(side)

def s2():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        r.sample_z(rays, 1)
    ms = r._march_begin(rays)
    r._march_refine(ms)
    main.wait_stream(side)

out = {}
for rep in range(2):
    out["S1_%d" % rep] = round(timed(s1), 4)
    out["S2_%d" % rep] = round(timed(s2), 4)
print(json.dumps(out))
