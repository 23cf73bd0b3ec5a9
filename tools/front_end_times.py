#!/usr/bin/env python3
"""Dev tool (GPU box): where the front end of a training step spends its time.  Times, with events on the launching streams,
the 128-proposal marching query, the 8-iteration secant chain and the hierarchical sampling chain each ALONE, and the two chains
running concurrently as the training step schedules them (trainer.compute_loss_fused).  Answers VERDICT r1 #4: is the secant chain
on the critical path?  ->  gpurun_out/front_end_times.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


ms = r._march_begin(rays)
out = {}
out["march_query_128_proposals_ms"] = timed(lambda: r._march_begin(rays))
out["secant_chain_alone_ms"] = timed(lambda: r._march_refine(ms))
out["sampling_chain_alone_ms"] = timed(lambda: r.sample_z(rays, 1))


def both():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        r.sample_z(rays, 1)
    r._march_refine(ms)
    main.wait_stream(side)


out["secant_and_sampling_concurrent_ms"] = timed(both)


def both_serial():
    r.sample_z(rays, 1)
    r._march_refine(ms)


out["secant_then_sampling_serial_ms"] = timed(both_serial)
# which chain ends last when they run concurrently?  events at the end of each chain, relative to a common start
torch.cuda.synchronize()
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(30)]
for e0, e_sec, e_smp in ev:          # host runs ahead (no synchronisation inside the loop), as in the training step
    e0.record(main)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        r.sample_z(rays, 1)
        e_smp.record(side)
    r._march_refine(ms)
    e_sec.record(main)
    main.wait_stream(side)
torch.cuda.synchronize()
out["concurrent_secant_chain_ends_at_ms"] = sum(e0.elapsed_time(e_sec) for e0, e_sec, _ in ev[5:]) / 25
out["concurrent_sampling_chain_ends_at_ms"] = sum(e0.elapsed_time(e_smp) for e0, _, e_smp in ev[5:]) / 25
out["note"] = ("the render forward needs the sampling result; the concurrent figure ~ max(chains) + contention: the secant chain is hidden "
               "whenever it is the shorter of the two")
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/front_end_times.json", "w"), indent=1)
