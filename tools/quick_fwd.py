"""Quick forward-only timing of renderer(rays) at config 2 (1024 rays x 32+32 samples). Dev tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import weightgen
from gpu_util import net_cfg
from oracle_util import RENDER_CFG
from endosurf_amd import EndoSurfRenderer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(0)
r = EndoSurfRenderer(dict(RENDER_CFG), net_cfg(True), device="cuda")
rays = torch.from_numpy(weightgen.make_rays(1, N)).cuda()
r.eval()
with torch.no_grad():
    for _ in range(3):
        ret = r(rays, iter_step=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ret = r(rays, iter_step=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
print(f"forward N={N}: {dt*1e3:.3f} ms/iter  {N/dt:.0f} rays/s  ({568.8e6*N/dt/1e12:.1f} TFLOP/s algorithmic)")
print("color mean", ret["color_map"].mean().item(), "wsum", ret["weights"].sum(-1).mean().item())
