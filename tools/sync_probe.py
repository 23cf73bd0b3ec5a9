import sys, time, json, torch, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
from bench import *
torch.cuda.set_device(0)
ctx = Ctx(torch.device("cuda", 0), 0, 1, False, False, "nccl", "plain")
loop = ReferenceLoop(ctx)
loop.renderer.engine.march_block = 0
for i in range(6): loop.train_step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20): loop.train_step(10 + i)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue()[:5000])
