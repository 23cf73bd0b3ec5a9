"""Dev probe: per-launch time of the small-batch SDF query (es_query_sdf, 16-point tiles) versus batch size."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path[:0] = [".", "tests"]
import weightgen  # noqa: E402
from endosurf_amd import _lib, params  # noqa: E402
from endosurf_amd._lib import es_points  # noqa: E402

lib = _lib.load()
_lib.check(lib.es_init(), "es_init")
for use_deform in (True, False):
    state = weightgen.make_state(3, "trained", use_deform)
    flat = torch.from_numpy(params.flatten_state(state)).cuda()
    weff = torch.zeros(lib.es_weff_floats(), device="cuda")
    packed = torch.zeros(lib.es_packed_floats(), device="cuda")
    _lib.check(lib.es_weightnorm_pack(_lib.ptr(flat), _lib.ptr(weff), _lib.ptr(packed), int(use_deform), _lib.stream_ptr()))
    for M in (16, 64, 256, 1024, 2048, 4096, 8192, 16384):
        x = (torch.rand(M, 3, device="cuda") * 1.6 - 0.8).contiguous()
        t = torch.rand(M, device="cuda")
        out = torch.empty(M, device="cuda")
        pts = es_points()
        pts.x, pts.t, pts.dirs, pts.rays, pts.z = _lib.ptr(x), _lib.ptr(t), None, None, None
        pts.mode, pts.t_scalar, pts.n_per_ray, pts.ldz, pts.M = 0, 0, 1, 1, M

        def go(n):
            for _ in range(n):
                lib.es_query_sdf(C.byref(pts), _lib.ptr(packed), _lib.ptr(weff), _lib.ptr(out), int(use_deform), _lib.stream_ptr())
        go(3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(50); e1.record(); torch.cuda.synchronize()
        print(f"deform={int(use_deform)} M={M:6d}  {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per launch")
