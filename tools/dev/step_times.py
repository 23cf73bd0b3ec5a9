#!/usr/bin/env python3
"""Dev tool (GPU box): host time vs total time of a training step (is the Python orchestration ever the bottleneck?).
usage: python tools/dev/step_times.py [config 2|3|4]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench as B
from endosurf_amd import EndoSurfRenderer
from endosurf_amd.trainer import SyntheticScene, Trainer

cfg = B.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 2]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
r = EndoSurfRenderer(B.render_cfg(cfg), dict(B.NET_CFG, use_deform=cfg["use_deform"]), device=dev)
r.engine.march_block = 0
tr = Trainer(r)
sc = SyntheticScene(dev, seed=1234)
batches = [sc.batch(cfg["rays"]) for _ in range(4)]
host, total = [], []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.update_learning_rate(i + 1); tr.train_step(batches[i % 4], i + 1)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
h, t = sorted(host[10:]), sorted(total[10:])
print(f"config {sys.argv[1] if len(sys.argv) > 1 else 2}: host median {h[len(h) // 2]:.2f} ms (max {h[-1]:.2f}), step median {t[len(t) // 2]:.2f} ms")
