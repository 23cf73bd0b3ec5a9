#!/usr/bin/env python3
"""Dev tool (GPU box): where the front end of a training step spends its time.  Times, with events on the launching streams,
the 128-proposal marching query, the 8-iteration secant chain and the hierarchical sampling chain each ALONE, and the two chains
running concurrently as the training step schedules them (trainer.compute_loss_fused).  Answers VERDICT r1 #4: is the secant chain
on the critical path?  ->  gpurun_out/front_end_times.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench as B
from endosurf_amd import EndoSurfRenderer
from endosurf_amd.trainer import SyntheticScene

dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg = B.CONFIGS[2]
r = EndoSurfRenderer(B.render_cfg(cfg), dict(B.NET_CFG), device=dev)
r.engine.march_block = 0
sc = SyntheticScene(dev, seed=1234)
rays = r._rays32(sc.batch(cfg["rays"])["rays"])
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)
r._weights()


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(n):
        fn()
    b.record(main)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


ms = r._march_begin(rays)
out = {}
out["march_query_128_proposals_ms"] = timed(lambda: r._march_begin(rays))
out["secant_chain_alone_ms"] = timed(lambda: r._march_refine(ms))
out["sampling_chain_alone_ms"] = timed(lambda: r.sample_z(rays, 1))


def both():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        r.sample_z(rays, 1, racing=True)          # as trainer.compute_loss_fused issues it (coarse query on 32-point tiles)
    r._march_refine(ms)
    main.wait_stream(side)


out["secant_and_sampling_concurrent_ms"] = timed(both)


def both_serial():
    r.sample_z(rays, 1)
    r._march_refine(ms)


out["secant_then_sampling_serial_ms"] = timed(both_serial)
# which chain ends last when they run concurrently?  events at the end of each chain, relative to a common start
torch.cuda.synchronize()
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(30)]
for e0, e_sec, e_smp in ev:          # host runs ahead (no synchronisation inside the loop), as in the training step
    e0.record(main)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        r.sample_z(rays, 1, racing=True)
        e_smp.record(side)
    r._march_refine(ms)
    e_sec.record(main)
    main.wait_stream(side)
torch.cuda.synchronize()
out["concurrent_secant_chain_ends_at_ms"] = sum(e0.elapsed_time(e_sec) for e0, e_sec, _ in ev[5:]) / 25
out["concurrent_sampling_chain_ends_at_ms"] = sum(e0.elapsed_time(e_smp) for e0, _, e_smp in ev[5:]) / 25
out["note"] = ("the render forward needs the sampling result; the concurrent figure ~ max(chains) + contention: the secant chain is hidden "
               "whenever it is the shorter of the two")
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/front_end_times.json", "w"), indent=1)


# ---- round 4: what would ONE grid per (sampling query i, secant iteration i) buy?  The launches of the merged schedule
# are issued with the kernels that exist -- the 8 192 new sample points and the 1 024 secant points of an iteration as ONE 9 216-point
# launch of the 16-point tiles (576 tiles; execution time does not depend on the data) -- in the order the merged step would issue them:
#   [coarse query (side) || secant #1 (main)]  ->  3 x [up-sample; secant points; query(9 216); merge; secant update]  ->  up-sample
#   ->  4 x [secant points; query(1 024); secant update]
# and timed end to end like the concurrent schedule above.  Results are meaningless, the timing is that of the merged schedule.
import ctypes as C
from endosurf_amd import _lib
eng = r.engine
weff, packed = r._weights()
weff = weff.detach()
N, n0, S, n_imp = cfg["rays"], cfg["n_samples"], cfg["n_samples"] + cfg["n_importance"], cfg["n_importance"] // 4
zc, zn = eng.empty(N, S), eng.empty(N, S)
eng.ray_setup(rays, None, n0, 2.0 / n0, 0, zc)
sdf_c = eng.query_sdf(eng.points(rays=rays, z=zc, n_per_ray=n0, ldz=S), weff, packed, True).view(N, n0)
sdf_a, src, z_new = eng.empty(N, S), eng.empty(N, S, dtype=torch.int32), eng.empty(N, n_imp)
xm, tm = (torch.rand(9216, 3, device=dev) - 0.5), torch.rand(9216, device=dev)
x1, t1 = xm[:1024].contiguous(), tm[:1024].contiguous()
xs, ts = eng.empty(N, 3), eng.empty(N)
st = eng.st


def secant_iter(x, t):
    _lib.check(eng.lib.es_secant_points(_lib.ptr(rays), _lib.ptr(ms["d_pred"]), N, _lib.ptr(xs), _lib.ptr(ts), st()), "sp")
    f = eng.query_sdf(eng.points(x=x, t=t), weff, packed, True)
    _lib.check(eng.lib.es_secant_update(_lib.ptr(f), N, 0.0, _lib.ptr(ms["state"]), _lib.ptr(ms["d_pred"]), st()), "su")
    return f


def upsample(i):
    _lib.check(eng.lib.es_upsample_step(_lib.ptr(rays), _lib.ptr(zc), S, _lib.ptr(sdf_c), n0, N, n0, n_imp, float(64 * 2 ** i), _lib.ptr(z_new),
                                        _lib.ptr(zn), S, _lib.ptr(src), st()), "up")


def merged():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        eng.ray_setup(rays, None, n0, 2.0 / n0, 0, zc)
        eng.query_sdf(eng.points(rays=rays, z=zc, n_per_ray=n0, ldz=S), weff, packed, True)
    secant_iter(x1, t1)
    main.wait_stream(side)
    for i in range(3):
        upsample(i)
        f = secant_iter(xm, tm)                         # ONE 9 216-point launch: 8 192 sample points + 1 024 secant points
        _lib.check(eng.lib.es_merge_sdf(_lib.ptr(sdf_c), n0, _lib.ptr(f), n_imp, _lib.ptr(src), S, N, n0, _lib.ptr(sdf_a), st()), "merge")
    upsample(3)
    for _ in range(4):
        secant_iter(x1, t1)


out["merged_grid_schedule_emulated_ms"] = timed(merged)
out["query16_9216_points_ms"] = timed(lambda: eng.query_sdf(eng.points(x=xm, t=tm), weff, packed, True))
out["query16_8192_points_ms"] = timed(lambda: eng.query_sdf(eng.points(x=xm[:8192].contiguous(), t=tm[:8192].contiguous()), weff, packed, True))
out["query16_1024_points_ms"] = timed(lambda: eng.query_sdf(eng.points(x=x1, t=t1), weff, packed, True))
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/front_end_times.json", "w"), indent=1)


# ---- round 5: the coarse 32 768-point query INSIDE the marching launch (one 163 840-point grid = 2 560 tiles = five
# full rounds of the chip's 512 workgroup slots instead of four rounds + a 512-tile launch racing the secant chain), leaving only
# 3 x (up-sample + 8 192-point query + merge) + the last up-sample on the sampling stream next to the secant chain.  Emulated with the
# kernels that exist: ONE query launch over 160 depths per ray (results meaningless, the timing is that of the fused grid), then the two
# chains side by side without the coarse query.
z160 = (torch.rand(N, 160, device=dev) * 2.0).contiguous()
sdf_c32 = eng.query_sdf(eng.points(rays=rays, z=zc, n_per_ray=n0, ldz=S), weff, packed, True).view(N, n0)


sdf_b2 = eng.empty(N, S)


def fused_grid():
    eng.query_sdf(eng.points(rays=rays, z=z160, n_per_ray=160, ldz=160), weff, packed, True)


zc0 = zc.clone()


def rest_of_sampling():
    """engine.sample_z without its ray set-up and coarse query (they ride in the fused grid)."""
    za, zb = zc, zn
    za.copy_(zc0)
    sd, ld, n = sdf_c32, n0, n0
    for i in range(4):
        _lib.check(eng.lib.es_upsample_step(_lib.ptr(rays), _lib.ptr(za), S, _lib.ptr(sd), ld, N, n, n_imp, float(64 * 2 ** i),
                                            _lib.ptr(z_new), _lib.ptr(zb), S, _lib.ptr(src), st()), "up")
        if i < 3:
            f = eng.query_sdf(eng.points(rays=rays, z=z_new, n_per_ray=n_imp, ldz=n_imp), weff, packed, True)
            dst = sdf_a if sd.data_ptr() != sdf_a.data_ptr() else sdf_b2
            _lib.check(eng.lib.es_merge_sdf(_lib.ptr(sd), ld, _lib.ptr(f), n_imp, _lib.ptr(src), S, N, n, _lib.ptr(dst), st()), "merge")
            sd, ld = dst, S
        za, zb = zb, za
        n += n_imp


def racing_without_coarse():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        rest_of_sampling()
    r._march_refine(ms)
    main.wait_stream(side)


def fused_front_end():
    fused_grid()
    racing_without_coarse()


def current_front_end():
    r._march_begin(rays)
    both()


out["fused_grid_163840_points_ms"] = timed(fused_grid)
out["sampling_chain_without_coarse_query_alone_ms"] = timed(rest_of_sampling)
out["racing_section_without_coarse_query_ms"] = timed(racing_without_coarse)
out["fused_front_end_emulated_ms"] = timed(fused_front_end)
out["current_front_end_ms"] = timed(current_front_end)
print(json.dumps({k: v for k, v in out.items() if "fused" in k or "without" in k or "current" in k}, indent=1))
json.dump(out, open("gpurun_out/front_end_times.json", "w"), indent=1)
