import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from endosurf_amd import EndoSurfRenderer
dev = torch.device("cuda", 0)
r = EndoSurfRenderer(dict(B.RENDER_CFG), B.NET_CFG, device=dev)
eng = r.engine
with torch.no_grad():
    weff, packed = r._weights()
    xs = torch.rand(1024, 3, device=dev) - 0.5
    ts = torch.rand(1024, device=dev)
    xb = torch.rand(131072, 3, device=dev) - 0.5
    tb = torch.rand(131072, device=dev)
    small = lambda: eng.query_sdf(eng.points(x=xs, t=ts), weff, packed, True)
    big = lambda: eng.query_sdf(eng.points(x=xb, t=tb), weff, packed, True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def timeit(f, n=5):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    def seq_small(): [small() for _ in range(16)]
    def par_small():
        with torch.cuda.stream(s1): [small() for _ in range(8)]
        with torch.cuda.stream(s2): [small() for _ in range(8)]
    def seq_mix(): big(); [small() for _ in range(8)]
    def par_mix():
        with torch.cuda.stream(s1): big()
        with torch.cuda.stream(s2): [small() for _ in range(8)]
    print("16 small sequential     %.3f ms" % timeit(seq_small))
    print("8+8 small on 2 streams  %.3f ms" % timeit(par_small))
    print("big + 8 small sequential %.3f ms" % timeit(seq_mix))
    print("big || 8 small 2 streams %.3f ms" % timeit(par_mix))
    print("big alone %.3f ms, 8 small alone %.3f ms" % (timeit(big), timeit(lambda: [small() for _ in range(8)])))
