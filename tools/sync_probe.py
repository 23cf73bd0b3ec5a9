import sys, time, json, torch
sys.path.insert(0, "/root/repo")
import bench
from bench import *
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
res = {}
def probe(name, schedule, flat, sync, march_block=0, n=15):
    ctx = Ctx(dev, 0, 1, False, False, "nccl", schedule)
    wl = Workload(ctx, 2)
    if not flat:
        from endosurf_amd.trainer import Trainer
        wl.trainer = Trainer(wl.renderer, schedule=schedule, flat_adam=False)
    eng = wl.eng
    eng.march_block = march_block
    def step(i):
        wl.step(i)
        if sync: torch.cuda.synchronize()
    for i in range(5): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): step(10 + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    t = kernel_timing(eng, step, 100, n, True)
    d = {k: v for k, v in t["per_step_ms"].items()}
    d["_ms_per_step"] = round(ms, 3)
    res[name] = d
    wl.close()
probe("fused_flat", "fused", True, 0)
probe("plain_flat", "plain", True, 0)
probe("plain_torchadam", "plain", False, 0)
probe("plain_torchadam_sync", "plain", False, 1)
probe("plain_torchadam_sync_exit", "plain", False, 1, 32)
keys = sorted(set(k for d in res.values() for k in d))
print("%-70s" % "kernel" + "".join("%22s" % n for n in res))
for k in keys:
    print("%-70s" % k[:70] + "".join("%22s" % d.get(k, "") for d in res.values()))
