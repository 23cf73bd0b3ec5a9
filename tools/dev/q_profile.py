#!/usr/bin/env python3
"""Dev tool: phase cycle stamps of the fp32 64-point SDF query kernel (library built with -DES_PROFILE_QUERY): the first block (first
round of the 512 workgroup slots) and the last block (last round) of the 131 072-point marching query, + their wall clocks (100 MHz)."""
import ctypes as C, sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
from gpu_util import renderer_for
r = renderer_for(24, "trained", True)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
x = torch.rand(M, 3, device="cuda") - 0.5; t = torch.rand(M, device="cuda")
for _ in range(3): r.sdf_observed(x, t)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); r.sdf_observed(x, t); b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b)
buf = (C.c_longlong * 192)()
r.engine.lib.es_debug_q_profile.restype = C.c_int
r.engine.lib.es_debug_q_profile(buf, 192)
v = list(buf)
names = ["load + deform encode", "deform layer 0", "deform layers 1-7", "deform tail (3 outputs) + x_c", "sdf encode + layer 0", "sdf layers 1-7", "sdf tail (1 output)"]
out = dict(points=M, tiles=M // 64, launch_ms=ms, blocks={})
for name, base in (("first block", 0), ("last block", 64)):
    w = v[base:base + 64]
    tot = w[7] - w[0]
    rows = {n: dict(cycles=w[i + 1] - w[i], share=round((w[i + 1] - w[i]) / tot, 4)) for i, n in enumerate(names)}
    out["blocks"][name] = dict(tile_cycles=tot, phases=rows, deform_layer_gemm_only_cycles=[w[20 + l] - w[10 + l] for l in range(1, 8)])
    print(name, "tile total cycles", tot)
    for n, d in rows.items():
        print(f"  {n:32s} {d['cycles']:8d}  {100 * d['share']:5.1f} %")
    print("  deform layer gemm-only cycles:", out["blocks"][name]["deform_layer_gemm_only_cycles"])
# wall clocks (100 MHz): start of the first block -> end of the last block = the launch as the tiles see it
w0, w0e, w1, w1e = v[128], v[128 + 7], v[128 + 32], v[128 + 32 + 7]
out["wall_100MHz"] = dict(first_block_us=(w0e - w0) / 100.0, last_block_us=(w1e - w1) / 100.0, first_start_to_last_end_us=(w1e - w0) / 100.0,
                          last_block_start_after_first_start_us=(w1 - w0) / 100.0)
print(json.dumps(out["wall_100MHz"]))
print("launch ms (events):", ms)
# every block's start / end (100 MHz wall clock): occupancy of the 512 workgroup slots over the launch
nb = M // 64
tb = (C.c_longlong * (2 * nb))()
r.engine.lib.es_debug_q_times.restype = C.c_int
r.engine.lib.es_debug_q_times(tb, 2 * nb)
import numpy as np
T = np.array(list(tb), dtype=np.int64).reshape(nb, 2).astype(np.float64) / 100.0      # us
t0, t1 = T[:, 0].min(), T[:, 1].max()
dur = T[:, 1] - T[:, 0]
span = t1 - t0
busy = dur.sum()                               # slot-microseconds of work
slots = 512
order = np.argsort(T[:, 0])
rounds = [dur[order[i * slots:(i + 1) * slots]] for i in range((nb + slots - 1) // slots)]
first_idle = np.sort(T[:, 1])[-slots]           # when the first slot runs out of tiles (the 512th-last end)
out["slots"] = dict(span_us=span, busy_slot_us=busy, mean_tile_us=float(dur.mean()), occupancy=busy / (span * slots),
                    tile_us_by_start_round=[dict(mean=float(x.mean()), min=float(x.min()), max=float(x.max())) for x in rounds],
                    drain_starts_us=float(first_idle - t0), drain_us=float(t1 - first_idle),
                    drain_idle_share=float(1.0 - dur.sum() / (span * slots)),
                    start_spread_first_round_us=float(np.sort(T[:, 0])[slots - 1] - t0))
print(json.dumps(out["slots"], indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/q_stamp.json", "w"), indent=1)
