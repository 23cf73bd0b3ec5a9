"""Dev tool: GPU memory must be flat over many training steps (reference cycles through autograd nodes would leak ~7 GB/step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
from endosurf_amd import EndoSurfRenderer
from endosurf_amd.trainer import SyntheticScene, Trainer
dev = torch.device("cuda", 0)
cfg = dict(B.CONFIGS[2])
r = EndoSurfRenderer(B.render_cfg(cfg), dict(B.NET_CFG, use_deform=cfg["use_deform"]), device=dev)
if os.environ.get("ES_SPLIT_BF16") == "1":
    r.engine.split_precision = True
tr = Trainer(r)
sc = SyntheticScene(dev, seed=1)
b = sc.batch(1024)
marks = []
for i in range(120):
    tr.update_learning_rate(i + 1)
    tr.train_step(b, i + 1)
    if i % 20 == 19:
        torch.cuda.synchronize()
        marks.append((i + 1, round(torch.cuda.memory_allocated() / 2**30, 3), round(torch.cuda.max_memory_allocated() / 2**30, 3), round(torch.cuda.memory_reserved() / 2**30, 3)))
print("(step, allocated GiB, peak GiB, reserved GiB):", marks)
assert marks[-1][1] <= marks[0][1] + 0.05 and marks[-1][3] <= marks[1][3] + 0.5, "memory grows"
print("ok")
