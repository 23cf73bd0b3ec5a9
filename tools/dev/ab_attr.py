#!/usr/bin/env python3
"""Dev tool (GPU box): A/B an Engine attribute inside ONE process: alternating blocks of training steps with the attribute off / on.
usage: python tools/dev/ab_attr.py <attribute> [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
from endosurf_amd import EndoSurfRenderer
from endosurf_amd.trainer import SyntheticScene, Trainer
attr, rounds = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
cfg = dict(B.CONFIGS[2])
r = EndoSurfRenderer(B.render_cfg(cfg), dict(B.NET_CFG, use_deform=True), device=dev)
r.engine.march_block = 0
tr = Trainer(r); sc = SyntheticScene(dev, seed=1); bs = [sc.batch(1024) for _ in range(4)]
step = 0
def run(n):
    global step
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step += 1; tr.update_learning_rate(step); tr.train_step(bs[step % 4], step)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(8)
for k in range(rounds):
    for v in (False, True):
        setattr(r.engine, attr, v); run(3)
        print(attr, v, round(run(40), 3), "ms/step")
