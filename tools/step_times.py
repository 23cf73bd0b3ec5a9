import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from endosurf_amd import EndoSurfRenderer
from endosurf_amd.trainer import SyntheticScene, Trainer
dev = torch.device("cuda", 0)
torch.manual_seed(0)
r = EndoSurfRenderer(dict(B.RENDER_CFG), B.NET_CFG, device=dev)
tr = Trainer(r)
sc = SyntheticScene(dev, seed=1234)
batches = [sc.batch(1024) for _ in range(4)]
ts = []
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.update_learning_rate(i + 1); tr.train_step(batches[i % 4], i + 1)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((round((t1 - t0) * 1e3, 1), round((t2 - t0) * 1e3, 1)))
print("(host ms, total ms) per step:", ts)
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
