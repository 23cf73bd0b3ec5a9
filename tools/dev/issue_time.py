"""Dev tool: host time to ISSUE a training step (no synchronisation) against its wall time: 3.4 ms of 13.4 / 11.8 ms (fp32 / split-precision,
marching early exit on): the step is GPU-bound with a 3-4x margin on the host.  usage: [ES_SPLIT_BF16=1] python tools/dev/issue_time.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench as B
from endosurf_amd import EndoSurfRenderer
from endosurf_amd.trainer import SyntheticScene, Trainer
dev = torch.device("cuda", 0)
cfg = dict(B.CONFIGS[2])
r = EndoSurfRenderer(B.render_cfg(cfg), dict(B.NET_CFG, use_deform=True), device=dev)
r.engine.split_precision = os.environ.get("ES_SPLIT_BF16") == "1"
tr = Trainer(r); sc = SyntheticScene(dev, seed=1); bs = [sc.batch(1024) for _ in range(4)]
for i in range(5):
    tr.update_learning_rate(i + 1); tr.train_step(bs[i % 4], i + 1)
torch.cuda.synchronize()
n = 30; t0 = time.perf_counter(); issue = 0.0
for i in range(n):
    a = time.perf_counter(); tr.update_learning_rate(i + 6); tr.train_step(bs[i % 4], i + 6); issue += time.perf_counter() - a
t_issue = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"split={r.engine.split_precision}: host issue {1e3 * issue / n:.2f} ms/step, wall {1e3 * t_all / n:.2f} ms/step, host loop done after {1e3 * t_issue / n:.2f} ms/step")
