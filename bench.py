#!/usr/bin/env python3
"""Headline benchmark: training rays/s of EndoSurf's renderer hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]

N > 1: one rank per GPU over RCCL.  Either launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``
(RANK / LOCAL_RANK / WORLD_SIZE in the environment) or directly as ``python bench.py --gpus N``, which re-executes itself under
torch.distributed.run on 127.0.0.1.

Workloads (BASELINE.json ``configs``; all synthetic rays / targets resident in HBM, reference-initialised weights, fp32):
  --config 2  (default, the configuration the metric is quoted on) base_pull.yml networks, 1024 rays x (32 coarse + 32 importance)
              samples per GPU, one FULL training step per "step": render (hierarchical sampling + fused MLP stack + compositing) +
              errorondepth + surface_neighbour_error (128-step ray marching + 8 secant steps) + loss + backward + Adam.
  --config 3  base_cut.yml networks, 2048 rays x (64 + 64) samples, same full training step (eikonal gradient included).
  --config 4  base_d1k1.yml networks (use_deform False), 1024 rays per GPU (4096-ray batch over 4 GPUs), full training step.
  --config 5  one 640x512 frame per step, forward only, hipGraph-captured 2048-ray chunks, frame rows split over the ranks.
``--mode forward`` times the renderer forward of the chosen training configuration instead.

The DEFAULT invocation (config 2, training) appends, after the headline and outside its timed region, ``cpu_baseline`` and ``extras``:
cfg2 forward-only, cfg2 in the opt-in split-precision mode, cfg3, cfg4 and one cfg5 frame -- each ``{ms_per_step, value,
roofline: {kernel, frac, peak, end_to_end}}`` from a few steps of a fresh renderer (``--no-extras`` / ``--headline-only`` skip them).  N > 1
lines carry ``ranks_seen_by_collective`` (an all-reduce of ones), ``ms_per_step_min / max`` over the ranks and ``allreduce_ms`` (HIP events
around the flat-bucket all-reduce).

``value`` is the DATA-INDEPENDENT step: ray marching evaluates all 128 proposals of every ray, as the reference does.  The early
exit at each ray's first sign change (bit-identical results, data-dependent saving) is reported next to it as
``config.with_early_exit``.  Weak scaling: every rank draws its own ray batch, one RCCL all-reduce of the 6.6 MB gradient bucket per
step.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

NET_CFG = dict(
    bound=1.0, use_deform=True,
    deform_network=dict(enc_pos_cfg=dict(enc_type="frequency", input_dim=3, multires=6),
                        enc_time_cfg=dict(enc_type="frequency", input_dim=1, multires=6), n_layers=9, hidden_dim=256, skips=[4], out_dim=3),
    sdf_network=dict(enc_pos_cfg=dict(enc_type="frequency", input_dim=3, multires=6), n_layers=9, hidden_dim=256, skips=[4], out_dim=257,
                     geometric_init=True, geometric_init_bias=0.8),
    color_network=dict(enc_pos_cfg=dict(enc_type="frequency", input_dim=3, multires=10),
                       enc_dir_cfg=dict(enc_type="frequency", input_dim=3, multires=4), n_layers=9, hidden_dim=256, skips=[4],
                       feat_dim=256, out_dim=3),
    deviation_network=dict(init_val=0.3))
CONFIGS = {
    2: dict(name="base_pull.yml", rays=1024, n_samples=32, n_importance=32, use_deform=True, mode="train"),
    3: dict(name="base_cut.yml (64+64 samples: BASELINE.json config 3)", rays=2048, n_samples=64, n_importance=64, use_deform=True, mode="train"),
    4: dict(name="base_d1k1.yml (use_deform False)", rays=1024, n_samples=32, n_importance=32, use_deform=False, mode="train"),
    5: dict(name="base_pull.yml", rays=640 * 512, n_samples=32, n_importance=32, use_deform=True, mode="frame"),
}
# per-point MACs of the three MLPs (SURVEY 8 / BASELINE.md 2)
MAC_D, MAC_S, MAC_C = 459520, 544512, 638208
PEAK_F32_MFMA = 157.3      # TFLOP/s, dense v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md)
PEAK_BF16_MFMA = 2500.0    # TFLOP/s, dense v_mfma_f32_32x32x16_bf16 (same table); used for the opt-in split-precision kernels only


def render_cfg(c):
    return dict(net_chunk=80000, anneal_end=50000, n_samples=c["n_samples"], n_importance=c["n_importance"], important_begin_iter=0,
                up_sample_steps=4, perturb=True)


def algorithmic_gflop_per_ray(c):
    """SURVEY 8d's closed forms with the executed deformation passes (3D per point instead of 4D: J d and J^T g_c, no full Jacobian)."""
    D = MAC_D if c["use_deform"] else 0
    S_c, S_i, steps = c["n_samples"], c["n_importance"], 4
    f_up = 2 * (S_c + S_i * (steps - 1) / steps) * (D + MAC_S)
    f_core = 2 * (S_c + S_i) * (3 * D + 2 * MAC_S + MAC_C)
    f_march = 2 * 128 * (D + MAC_S)
    return dict(upsample=f_up / 1e9, render_core_forward=f_core / 1e9, ray_marching=f_march / 1e9, train_step=(f_up + 3 * f_core + f_march) / 1e9)


def kernel_macs(name, use_deform, executed=True):
    """MACs per point of each timed kernel.  The deformation network runs as value + JVP (J d) rows (2D), one VJP sweep (J^T g_c, D) and,
    in the backward, one tangent sweep (J gbar_o, D); SDF 2S (value + reverse / tangent + reverse), colour C; weight gradients the same
    again; the SDF queries D + S.  Launches whose tiles may exit early are not counted as work.

    ``executed`` (default): the MACs the kernels ISSUE.  The SDF network's last layer is [257 x 256]; a query needs only its sdf row
    (query.hip: one 256-MAC dot product instead of 65 792 MACs), the analytic reverse sweep of k_sdf_fwd starts from that one row, the
    tangent sweep of k_sdf_bwd ends at tau_8 and the g_c-path weight gradient of the last layer is a column sum: the nominal SURVEY 8d
    figures (``executed=False``: D, S, C per pass) overstate these kernels by 65 536 / 65 792 MACs per point (6.5 % of a query)."""
    D = MAC_D if use_deform else 0
    S8 = 257 * 256                     # MACs of the SDF network's last layer (sdf row + 256 feature rows)
    cut = {"k_query_sdf": S8 - 256, "k_query_sdf16": S8 - 256, "k_query_sdf_x3": S8 - 256,      # sdf row only
           "k_sdf_fwd": S8 - 256, "k_sdf_fwd_x3": S8 - 256,      # reverse sweep: rho_7 = softplus'(z_7) . W8[0, :], 256 multiplies
           "k_sdf_bwd": S8,                                      # tangent sweep stops at tau_8; reverse: SR8F (65 536) + the sdf row (256)
           "k_wgrad[sdf]": S8 - 256, "k_wgrad_x3[sdf]": S8 - 256}      # g_c-path pair of the last layer: column sums of tau_8
    nominal = {"k_query_sdf": D + MAC_S, "k_query_sdf16": D + MAC_S, "k_query_sdf_x3": D + MAC_S, "k_deform_fwd": 2 * MAC_D, "k_deform_vjp": MAC_D,
               "k_sdf_fwd": 2 * MAC_S, "k_color_fwd": MAC_C, "k_color_bwd": MAC_C, "k_sdf_bwd": 2 * MAC_S, "k_deform_tan": MAC_D,
               "k_deform_bwd": 2 * MAC_D, "k_wgrad[deform]": 3 * MAC_D, "k_wgrad[sdf]": 2 * MAC_S, "k_wgrad[color]": MAC_C,
               "k_wgrad_x3[deform]": 3 * MAC_D, "k_wgrad_x3[sdf]": 2 * MAC_S, "k_wgrad_x3[color]": MAC_C,
               "k_deform_fwd_x3": 2 * MAC_D, "k_deform_vjp_x3": MAC_D, "k_sdf_fwd_x3": 2 * MAC_S, "k_color_fwd_x3": MAC_C,
               "k_deform_tan_x3": MAC_D, "k_deform_bwd_x3": 2 * MAC_D, "k_color_bwd_x3": MAC_C, "k_sdf_bwd_x3": 2 * MAC_S}.get(name)
    if nominal is None:
        return None
    if name.endswith("_x3") and name[:-3] in cut:
        cut[name] = cut[name[:-3]]
    return nominal - (cut.get(name, 0) if executed else 0)


def cpu_baseline(n_rays=1024, timed_steps=5, threads=16, extra_256=True):
    """The oracle (CPU restatement of the reference op sequence, torch-CPU fp32 + autograd) timed on this box's host cores on a bounded
    sample of the same workload, as SURVEY 8d asks: full training steps at ``n_rays`` rays -- the headline's own batch (1 024 rays x 64
    samples) -- one warm-up + ``timed_steps`` (>= 5) timed steps, each timed on its own; ``value`` is taken at the MEDIAN step, min / max are in
    the record (~11 s per step: ~70 s of the driver's run).  BASELINE config 1's 256-ray batch rides along as an extra.  16 intra-op threads:
    at these tensor sizes (8 192 points x 256 features per GEMM) torch-CPU is fastest there (measured 8/16/32/64/128 threads on the 2 x
    64-core host: 180 / 213 / 147 / 73 / 26 rays/s); ``cores`` reports the threads actually used."""
    import torch
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, max(1, os.cpu_count() or threads)))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import numpy as np
    import weightgen
    from oracle import endosurf_oracle as O
    state = weightgen.make_state(0, "init", True)
    params = {k: torch.tensor(v, requires_grad=True) for k, v in state.items()}
    R = O.OracleRenderer(O.OracleNet(params, True), render_cfg(CONFIGS[2]))
    opt = torch.optim.Adam(list(params.values()), lr=5e-4)
    rays = torch.from_numpy(weightgen.make_rays(1, n_rays))
    tg = {k: torch.from_numpy(v) for k, v in weightgen.make_targets(2, n_rays).items()}
    batch = dict(rays=rays, **tg)
    rng = np.random.default_rng(0)

    def step():
        opt.zero_grad()
        u = torch.from_numpy(rng.uniform(size=(n_rays, 1)).astype(np.float32))
        un = torch.from_numpy(rng.uniform(size=(n_rays, 3)).astype(np.float32))
        loss, _, _ = O.train_loss(R, batch, 1, u, un)
        loss.backward()
        opt.step()
    step()
    ts = []
    for _ in range(max(1, int(timed_steps))):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[len(ts) // 2]
    cores = torch.get_num_threads()
    torch.set_num_threads(prev_threads)
    extra = None
    if extra_256 and n_rays != 256:       # round 3 reported this sample size (BASELINE config 1): kept as an extra so the series stays comparable
        e = cpu_baseline(256, timed_steps=3, threads=threads, extra_256=False)
        extra = dict(value=e["value"], n_rays=256, sample=e["sample"])
    what = ("the batch the headline is measured on" if n_rays == 1024 else
            ("BASELINE config 1: the 256-ray batch" if n_rays == 256 else "a sample of the 1024-ray batch"))
    return dict(value=n_rays / med, unit="rays/s", cores=cores, kind="port", steps=len(ts), s_per_step_median=med, s_per_step_min=min(ts),
                s_per_step_max=max(ts), value_min=n_rays / max(ts), value_max=n_rays / min(ts),
                sample=f"1 warm-up + {len(ts)} timed steps, median: full training steps (config 2 networks and loss) of the CPU oracle at {n_rays} "
                       f"rays x 64 samples ({what}), torch-CPU fp32 + autograd, {med:.2f} s/step (min {min(ts):.2f}, max {max(ts):.2f})",
                config=dict(n_rays=n_rays, samples_per_ray=64, threads=cores), at_256_rays=extra,
                reference_in_build_container="profiles/reference_cpu.json: the reference itself (imported unmodified), 8 vCPU build container")


def pmc_info(symbol):
    """Counters of a kernel symbol from the newest committed rocprofv3 PMC summary that has it (profiles/*_pmc_summary.json; separate
    --pmc passes, tools/pmc_summary.py).  A summary row is one (instantiation, grid size) of the symbol with its launch count; a symbol
    may have several (k_query_sdf: the 131 072-point marching query on 64-point tiles and the 32 768-point coarse query on 32-point
    tiles).  Every figure returned is PER SINGLE LAUNCH, as a launch-weighted mean over the rows -- the same mix ``avg_launch_ms`` of the
    live timers averages over (each step launches every row's kernel as often as the profiled steps did): HBM bytes (FETCH_SIZE x 2 +
    WRITE_SIZE), cycles, cycle-weighted MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)); ``rows`` keeps the
    per-(instantiation, grid) figures.  None if no summary has the symbol."""
    import glob
    # newest summary that has the symbol (by name: the round tags sort; mtimes are meaningless on a fresh copy of the tree)
    rows, used = [], None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc_summary.json")), reverse=True):
        rows = [r for r in json.load(open(f)) if r.get("logical", r["kernel"].split("<")[0]) == symbol and r.get("hbm_bytes") == r.get("hbm_bytes")]
        if rows:
            used = f
            break
    if not rows:
        return None
    n = sum(r["launches"] for r in rows)
    cyc = sum(r["cycles"] * r["launches"] for r in rows)
    return dict(hbm_bytes=sum(r["hbm_bytes"] * r["launches"] for r in rows) / n, cycles=cyc / n,
                mfma_util=sum(r["mfma_util"] * r["cycles"] * r["launches"] for r in rows) / cyc if cyc else None,
                launches=n, source=os.path.basename(used),
                rows=[dict(kernel=r["kernel"], grid_threads=r["grid_threads"], launches=r["launches"], cycles=r["cycles"],
                           mfma_util=r["mfma_util"], hbm_bytes=r["hbm_bytes"]) for r in rows])


def pmc_rates(pm, avg_launch_ms):
    """(hbm_gbps, clock_ghz) that the committed counters mean at the launch duration measured live, or (None, None) with a reason when
    either falls outside what the part can do (shader clock 1.0 - 2.6 GHz, HBM <= 8 000 GB/s): a counter summary of another build or
    another launch mix must not print an impossible figure next to the measured ones."""
    gbps = pm["hbm_bytes"] / (avg_launch_ms * 1e-3) / 1e9
    ghz = pm["cycles"] / (avg_launch_ms * 1e-3) / 1e9
    if not (1.0 <= ghz <= 2.6) or gbps > 8000.0:
        return None, None, ("withheld: the committed counters (%s) and the live launch time disagree (%.2f GHz, %.0f GB/s): different "
                            "build or launch mix" % (pm["source"], ghz, gbps))
    return gbps, ghz, None


# ---- algorithmic HBM bytes of the stream-heavy kernels -----------------------------------------------------------------------
# Per POINT and body, from the workspace layout (csrc/workspace.h) and the loads / stores the tile bodies issue (point_fwd_bodies.h,
# point_bwd_bodies.h, wgrad.hip): every saved stack is fp32 [layers][M][256] = 1 024 B per layer and point ("K" below); a body's figure is
# each stack it must write once + each stack it must read once (+ the re-reads its two-sweep STRUCTURE implies, named below).  Small
# per-point vectors (3 - 64 floats) are included where they exceed 100 B.
_K = 1024
HBM_BYTES_PER_POINT = {
    # deformation value + tangent rows (2 rows per point): u_1..u_8 (16 K), encoding rows U0 (2 x 64 floats), mask words (8 x 32 B), x_c | J d
    "deform_fwd": 16 * _K + 512 + 256 + 24,
    # VJP sweep: r_0..r_7 (8 K), mask words (8 x 64 B: a 64-point tile reads both 32-point producers' words), g_c in, g_o | curvature out
    "deform_vjp": 8 * _K + 512 + 36,
    # SDF value pass + reverse sweep: s_1..s_8 written (8 K) and s_1..s_7 read back by the reverse sweep (7 K: the two-sweep structure),
    # rho_0..rho_7 (8 K), enc(x_c) and its adjoint (2 x 64 floats), the 256 geometry features (1 K, colour points only)
    "sdf_fwd": 8 * _K + 7 * _K + 8 * _K + 512 + 28,
    "sdf_fwd_feat": _K,
    # colour: h_1..h_8 (8 K), features read at layer 0 and again at the skip layer (2 K), the 93-wide small input written (128 floats) and
    # re-read at the skip layer (96 floats), mask words (8 x 32 B), x_c | g_c | J d in, rgb out
    "color_fwd": 8 * _K + 2 * _K + 512 + 384 + 256 + 48,
    # colour reverse sweep: y_0..y_7 (8 K) + y_8 (4 floats), featbar written at layer 0 and read-modified-written at the skip layer (3 K),
    # the small part's adjoint likewise (3 x 128 floats), mask words, three 3-vectors out
    "color_bwd": 8 * _K + 16 + 3 * _K + 1536 + 256 + 60,
    # deformation tangent sweep along gbar_o: tau_1..tau_8 (8 K) + its encoding tangent (64 floats), mask words, gbar_o in, J gbar_o out
    "deform_tan": 8 * _K + 256 + 512 + 24,
    # SDF backward = tangent sweep (reads s_l, rho_l; writes tau_l, zeta_l: 4 x 8 K) + reverse sweep (reads s_l AGAIN and zeta_l, writes
    # zbar_l in place: 3 x 8 K): the 7 x 8 K = 56 KiB floor of the two-sweep structure (DEAD_ENDS A9); tau_0, adj_eps (2 x 64 floats)
    "sdf_bwd": 7 * 8 * _K + 512 + 60,
    "sdf_bwd_feat": _K,
    # deformation reverse sweep (2 rows per point): a_0..a_7 (16 K) + a_8 (2 x 4 floats), mask words, seeds in
    "deform_bwd": 16 * _K + 32 + 256 + 24,
    # weight gradients: every (layer input X, adjoint dA) pair read ONCE.  deform: value + tangent rows (u 16 K + U0 512 | a 16 K + a_8 32)
    # and the VJP path's pair (tau 8 K + tau_0 256 | r 8 K); sdf: (s 8 K + enc 256 | zbar 8 K) and (tau 8 K + tau_0 256 | rho 8 K) + featbar
    # (1 K, colour points); colour: (h 8 K + features 2 x 1 K + small part 2 x 96 floats | y 8 K + y_8 16)
    "wgrad_deform": 16 * _K + 512 + 16 * _K + 32 + 8 * _K + 256 + 8 * _K + 12,
    "wgrad_sdf": 8 * _K + 256 + 8 * _K + 8 * _K + 256 + 8 * _K,
    "wgrad_sdf_feat": _K,
    "wgrad_color": 8 * _K + 2 * _K + 768 + 8 * _K + 16,
}
_WG_NOTE = ("each [256 x 128] task reads its dA rows in full, so the two k-block tasks of a row chunk read dA twice: once more than the "
            "algorithmic count (+8 KiB per point and pair).  Their block ids are 8 apart (same XCD, back to back: wgrad.hip wg_decode) so that the "
            "second read can hit that XCD's L2")


def hbm_accounting(P, T):
    """Algorithmic HBM bytes per launch of the headline step's stream-heavy symbols (P = rays x samples colour points, T = the colour-less
    auxiliary points riding in the same launches) beside the committed PMC traffic of the same symbol, and their ratio."""
    B = HBM_BYTES_PER_POINT
    tail_fwd = B["sdf_fwd"] + B["deform_vjp"]          # the tail's [sdf + vjp] tiles ride in the main deformation launch (point_fwd.hip)
    tail_bwd = B["deform_tan"] + B["sdf_bwd"]          # the tail's [tan + sdf_bwd] tiles ride in the main deformation reverse sweep
    comp = {
        "k_deform_fwd": (P * B["deform_fwd"] + T * tail_fwd, "P x deform_fwd + T x (sdf_fwd + deform_vjp): the tail's dependent stages ride in this launch", None),
        "k_sdf_fwd": (P * (B["sdf_fwd"] + B["sdf_fwd_feat"]), "P x sdf_fwd (incl. the geometry features)", None),
        "k_color_fwd": (P * B["color_fwd"], "P x color_fwd", None),
        "k_deform_vjp": (P * B["deform_vjp"], "P x deform_vjp", None),
        "k_color_bwd": (P * B["color_bwd"], "P x color_bwd", None),
        "k_deform_tan": (P * B["deform_tan"], "P x deform_tan", None),
        "k_sdf_bwd": (P * (B["sdf_bwd"] + B["sdf_bwd_feat"]), "P x sdf_bwd (incl. featbar)", None),
        "k_deform_bwd": (P * B["deform_bwd"] + T * tail_bwd, "P x deform_bwd + T x (deform_tan + sdf_bwd): the tail's dependent stages ride in this launch", None),
        "k_wgrad[deform]": ((P + T) * B["wgrad_deform"], "(P + T) x wgrad_deform", _WG_NOTE + "; the row-major deformation stacks mostly do"),
        "k_wgrad[sdf]": ((P + T) * B["wgrad_sdf"] + P * B["wgrad_sdf_feat"], "(P + T) x wgrad_sdf + P x featbar",
                         _WG_NOTE + "; for the fragment-ordered SDF stacks it does not: both dA stacks (zbar, rho) come from HBM twice, "
                         "+16 KiB per point = the whole excess"),
        "k_wgrad[color]": (P * B["wgrad_color"], "P x wgrad_color", _WG_NOTE + "; here it does not: +8 KiB per point = the whole excess"),
    }
    out = []
    for sym, (alg, what, note) in comp.items():
        pm = pmc_info(sym)
        rows = [r for r in (pm["rows"] if pm else []) if r["grid_threads"] >= 131072]      # the main launch of the symbol (not its small pieces)
        meas = (sum(r["hbm_bytes"] * r["launches"] for r in rows) / sum(r["launches"] for r in rows)) if rows else None
        out.append(dict(kernel=sym, algorithmic_bytes_per_launch=int(alg), composition=what,
                        pmc_hbm_bytes_per_launch=meas, pmc_source=pm["source"] if pm else None,
                        ratio=round(meas / alg, 3) if meas else None, per_point_algorithmic_bytes=round(alg / (P + T if "P + T" in what or "T x" in what else P), 1),
                        explanation=note))
    return out


def pmc_traffic(symbol):
    p = pmc_info(symbol)
    return (p["hbm_bytes"], p["source"]) if p else None


def kernel_timing(eng, step, first_step, n_steps, use_deform, record=True, split=False):
    """A few extra steps of the SAME workload with the library's HIP-event timers on (events recorded on the launch stream
    around every chain / query / weight-gradient kernel).  Kept out of the headline region so the events do not perturb ``value``."""
    import torch
    if record:
        eng.timing_enable(True)
        eng.timing_drain()
    for i in range(n_steps):
        step(first_step + i)
    torch.cuda.synchronize()
    if not record:
        return {}
    rec = eng.timing_drain()
    eng.timing_enable(False)
    sym = {}              # symbol -> [total ms, launches, total executed flops, total nominal flops]
    groups = {}           # (symbol, rows) -> [total ms, launches]
    for name, rows, ms in rec:
        macs, nom = kernel_macs(name, use_deform), kernel_macs(name, use_deform, executed=False)
        s = sym.setdefault(name, [0.0, 0, 0.0, 0.0])
        s[0] += ms; s[1] += 1; s[2] += 2.0 * macs * rows if macs else 0.0; s[3] += 2.0 * nom * rows if nom else 0.0
        g = groups.setdefault((name, rows), [0.0, 0])
        g[0] += ms; g[1] += 1
    out = {"per_step_ms": {k: round(v[0] / n_steps, 4) for k, v in sorted(sym.items(), key=lambda kv: -kv[1][0])}}
    out["flops_per_step"] = sum(v[2] for v in sym.values()) / n_steps
    out["nominal_flops_per_step"] = sum(v[3] for v in sym.values()) / n_steps
    # which matrix pipe a timed kernel's GEMMs run on: the split-precision kernels (suffix _x3) issue SIX bf16 MFMA products per
    # fp32-equivalent MAC on v_mfma_f32_32x32x16_bf16, everything else runs v_mfma_f32_32x32x2_f32 / 16x16x4_f32
    pipes = {}
    for k, v in sym.items():
        if not v[2]:
            continue
        p = pipes.setdefault("bf16" if "_x3" in k else "fp32", [0.0, 0.0])
        p[0] += v[0]; p[1] += v[2]
    out["pipes"] = {}
    for name, (tot, fl) in pipes.items():
        teq = fl / (tot * 1e-3) / 1e12
        if name == "fp32":
            out["pipes"][name] = dict(ms=round(tot / n_steps, 4), tflops=round(teq, 2), frac_of_157_3=round(teq / PEAK_F32_MFMA, 4))
        else:
            out["pipes"][name] = dict(ms=round(tot / n_steps, 4), fp32_equivalent_tflops=round(teq, 2), bf16_tflops=round(6.0 * teq, 1),
                                      frac_of_2500=round(6.0 * teq / PEAK_BF16_MFMA, 4))
    cand = [(v[0], k) for k, v in sym.items() if v[2] > 0]
    if split and any("_x3" in k for _, k in cand):
        # the mode's own kernels: the dominant symbol is chosen among the bf16-pipe kernels (the fp32 latency kernels the mode leaves
        # untouched would otherwise describe it: VERDICT r3 weak #9); roofline.pipes carries both pipes' totals
        cand = [c for c in cand if "_x3" in c[1]]
    if cand:
        _, name = max(cand)             # dominant kernel = the SYMBOL with the largest total time (as rocprofv3 --stats ranks them)
        tot, cnt, fl, fl_nom = sym[name]
        out["dominant"] = dict(kernel=name, avg_launch_ms=tot / cnt, launches=cnt, flops_per_launch=fl / cnt, tflops=fl / (tot * 1e-3) / 1e12,
                               nominal_flops_per_launch=fl_nom / cnt, nominal_tflops=fl_nom / (tot * 1e-3) / 1e12,
                               share_of_timed_kernel_time=tot / sum(v[0] for v in sym.values()))
    out["symbols"] = [dict(kernel=k, launches_per_step=v[1] / n_steps, ms_per_step=round(v[0] / n_steps, 4),
                           tflops=round(v[2] / (v[0] * 1e-3) / 1e12, 2) if v[2] else None) for k, v in sorted(sym.items(), key=lambda kv: -kv[1][0])]
    out["launch_groups"] = [dict(kernel=n, points=r, launches=c, avg_ms=round(t / c, 4),
                                 tflops=round(2.0 * kernel_macs(n, use_deform) * r / (t / c * 1e-3) / 1e12, 2) if kernel_macs(n, use_deform) else None)
                            for (n, r), (t, c) in sorted(groups.items(), key=lambda kv: -kv[1][0])]
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Ctx:
    """What every workload of one bench.py process shares: the rank's device and the process group."""

    def __init__(self, dev, rank, world, dist_on, force_dist, backend, schedule, overlap_allreduce=False):
        self.dev, self.rank, self.world, self.dist_on, self.force_dist, self.backend, self.schedule = dev, rank, world, dist_on, force_dist, backend, schedule
        self.overlap_allreduce = bool(overlap_allreduce)

    def barrier(self):
        import torch
        if self.dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, x):
        import torch
        if not self.dist_on:
            return x
        t = torch.tensor([x], device=self.dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def gather_over_ranks(self, x):
        import torch
        if not self.dist_on:
            return [x]
        t = torch.tensor([x], device=self.dev, dtype=torch.float64)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        torch.distributed.all_gather(out, t)
        return [float(o.item()) for o in out]


    def gather_obj(self, x):
        import torch
        if not self.dist_on:
            return [x]
        out = [None] * self.world
        torch.distributed.all_gather_object(out, x)
        return out


class Workload:
    """One BASELINE.json configuration on this rank: renderer + trainer + resident synthetic batches, ``step(i)`` = one pass of the hot
    path (a full training step, a forward, or one 640x512 frame)."""

    def __init__(self, ctx, config_id, mode=None, split=False, rays=None, chunk=2048, graph=False, frame_graph=True):
        import torch
        from endosurf_amd import EndoSurfRenderer, parallel
        from endosurf_amd.trainer import SyntheticScene, Trainer
        cfg = dict(CONFIGS[config_id])
        mode = mode or cfg["mode"]
        if mode == "frame" and cfg["mode"] != "frame":
            cfg = dict(CONFIGS[5], use_deform=cfg["use_deform"], n_samples=cfg["n_samples"], n_importance=cfg["n_importance"], name=cfg["name"])
        if rays:
            cfg["rays"] = rays
        self.ctx, self.cfg, self.mode, self.config_id, self.split, self.chunk, self.rays_override = ctx, cfg, mode, config_id, bool(split), chunk, rays
        self.graph, self.frame_graph = bool(graph) and mode == "train", frame_graph
        self.graph_probe = False          # set by main() for the headline workload of an N > 1 run (or --graph-probe)
        torch.manual_seed(0)
        self.renderer = EndoSurfRenderer(render_cfg(cfg), dict(NET_CFG, use_deform=cfg["use_deform"]), device=ctx.dev)
        if split:
            self.renderer.engine.split_precision = True
        self.trainer = Trainer(self.renderer, data_parallel=ctx.dist_on, schedule=ctx.schedule, force_collective=ctx.force_dist,
                               overlap_allreduce=ctx.overlap_allreduce)
        parallel.broadcast_parameters(self.trainer.params)
        self.scene = SyntheticScene(ctx.dev, seed=1234 + ctx.rank)
        self.eng = self.renderer.engine
        self.S = cfg["n_samples"] + cfg["n_importance"]
        if mode == "frame":
            # cfg5: one 640x512 frame per step, forward only, fixed 2048-ray chunks through one captured hipGraph; the frame's rows
            # are split across the ranks and every step ends with the image on rank 0 (parallel.gather_frame: one all-gather of the
            # packed colour | depth | normal slabs), as the reference's eval loop ends in one image (trainer_endosurf.py:221-240)
            self.H, self.W = 512, 640
            self.row0, self.rows = parallel.frame_rows(self.H, ctx.rank, ctx.world)
            self.frame_rays = self.scene.frame(H=self.H, W=self.W, t=0.5, row0=self.row0, rows=self.rows)
            self.n_rays = self.rows * self.W             # THIS rank's rays per step (the job renders H x W per step)
            self.rays_per_step_job = self.H * self.W
        else:
            self.n_rays = cfg["rays"]
            self.rays_per_step_job = ctx.world * self.n_rays
            self.batches = [self.scene.batch(self.n_rays) for _ in range(4)]      # resident in HBM before the timed region
        self.use_graph = self.graph

    def step(self, i):
        import torch
        from endosurf_amd import parallel
        if self.mode == "frame":
            out = self.renderer.render_frames(self.frame_rays, iter_step=1, ray_chunk=self.chunk, perturb_overwrite=False, use_graph=self.frame_graph)
            if self.ctx.world > 1:
                out = parallel.gather_frame(out, self.H, self.W)
            return out
        b = self.batches[i % len(self.batches)]
        if self.mode == "train":
            if self.use_graph:
                self.trainer.train_step_graph(b, i + 1)
            else:
                self.trainer.update_learning_rate(i + 1)
                self.trainer.train_step(b, i + 1)
        else:
            with torch.no_grad():
                self.renderer(b["rays"], iter_step=i + 1)

    def timed(self, first, n):
        """EXACTLY n steps bracketed by barrier + synchronize on both sides; -> (max over ranks, this rank's) seconds."""
        self.ctx.barrier()
        t0 = time.perf_counter()
        for i in range(n):
            self.step(first + i)
        self.ctx.barrier()
        dt = time.perf_counter() - t0
        return self.ctx.max_over_ranks(dt), dt

    def host_issue_ms(self, first, n=5):
        """CPU time to ENQUEUE one step (no synchronisation inside; the queue is drained before each sample so that no launch call blocks
        on a full queue): what the host must sustain per step for the GPU never to wait for it.  Median of n."""
        import torch
        ts = []
        for i in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.step(first + i)
            ts.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
        return sorted(ts)[len(ts) // 2]

    def probe_graph(self, first, steps, dt_eager, dt_eager_local, issue_ms):
        """-> dict(ms_per_step (max over ranks), per-rank gpu_idle_ms estimate, the fallback decision); see measure()."""
        ctx = self.ctx
        out = dict(steps=steps)
        try:
            self.use_graph = True
            for i in range(4):          # two eager initialisation steps, the capture + first replay, one more replay
                self.step(first + i)
            dtg, dtg_local = self.timed(first + 4, steps)
        except Exception as e:          # (a failed capture must not cost the eager line)
            self.use_graph = False
            return dict(error="%s: %s" % (type(e).__name__, str(e)[:300]), use_graph=False)
        self.use_graph = False
        eager_ms, graph_ms = dt_eager / steps * 1e3, dtg / steps * 1e3
        issue_max = ctx.max_over_ranks(issue_ms)
        host_bound = issue_max > 0.8 * eager_ms
        use = bool(host_bound and graph_ms < eager_ms)
        out.update(ms_per_step=graph_ms, eager_ms_per_step=eager_ms, host_issue_ms_max=issue_max, host_bound=bool(host_bound), use_graph=use,
                   gpu_idle_ms=max(0.0, (dt_eager_local - dtg_local) / steps * 1e3), dt=dtg, dt_local=dtg_local,
                   rule="headline = the replayed step iff rank-max host_issue_ms > 0.8 x eager ms_per_step and the replayed step is faster")
        return out

    def measure(self, warmup, steps, timing_steps=3, early_exit_extra=True):
        """warm-up, the timed region, (train) the same step with the marching early exit, then a few instrumented steps."""
        eng, mode = self.eng, self.mode
        march_block = eng.march_block
        if mode == "train":
            eng.march_block = 0          # headline: the data-independent step (every ray's 128 marching proposals, like the reference)
        for i in range(warmup):
            self.step(i)
        dt, dt_local = self.timed(warmup, steps)
        nxt = warmup + steps
        # the same step with ray marching's early exit (blocks of 32 proposals; tiles whose rays have all passed their first sign change
        # return at once; results bit-identical): data-dependent, reported as an extra
        extra = None
        if mode == "train" and march_block and early_exit_extra and not self.use_graph:      # (a captured step keeps the marching mode it was captured with)
            eng.march_block = march_block
            self.step(nxt)
            dte, _ = self.timed(nxt + 1, steps)
            nxt += 1 + steps
            extra = dict(ms_per_step=dte / steps * 1e3, value=self.rays_per_step_job * steps / dte, steps=steps, block=march_block,
                         note="results bit-identical to the headline step; on this synthetic init-weight scene every ray's first sign change "
                              "falls in the first block of 32 proposals (best case)")
            eng.march_block = 0
        issue_ms = self.host_issue_ms(nxt) if mode != "frame" else None
        nxt += 5
        # N > 1 (cold-run kit): the same step replayed from the whole-step hipGraph -- a step that needs NO host work between its first
        # and its last launch.  Its time is this rank's GPU-bound step time (on one GPU it is within 1 % of the eager step), so
        # eager - graph is the time per step the GPU spent waiting for this rank's host; and when the host is the limiter (rank-max
        # host_issue_ms > 0.8 x ms_per_step) and the replayed step is faster, the replayed step IS the headline (config says so).
        probe = None
        if mode == "train" and self.graph_probe and not self.use_graph:
            probe = self.probe_graph(nxt, steps, dt, dt_local, issue_ms)
            nxt += 4 + steps
            if probe.get("use_graph"):
                dt, dt_local = probe["dt"], probe["dt_local"]
        self.use_graph = False         # (events cannot be recorded inside a captured graph: the per-kernel timers run on eager steps)
        rec = self.ctx.rank == 0       # every rank runs the instrumented steps (they contain the gradient all-reduce); rank 0 records
        if mode == "frame":
            import torch

            def eager(_i):
                with torch.no_grad():
                    self.renderer(self.frame_rays.reshape(-1, 9)[:self.chunk], iter_step=1, perturb_overwrite=False)
            eager(0)
            timing = kernel_timing(eng, eager, 0, 4, self.cfg["use_deform"], record=rec, split=self.split)
            flops_per_step = timing.get("flops_per_step", 0.0) * (self.rays_per_step_job / self.ctx.world / self.chunk)
        else:
            timing = kernel_timing(eng, self.step, nxt + 64, timing_steps, self.cfg["use_deform"], record=rec, split=self.split)
            flops_per_step = timing.get("flops_per_step", 0.0)
        eng.march_block = march_block
        self.use_graph = self.graph
        per = timing.get("per_step_ms")
        return dict(dt=dt, dt_local=dt_local, steps=steps, warmup=warmup, ms=dt / steps * 1e3, value=self.rays_per_step_job * steps / dt,
                    with_early_exit=extra, timing=timing, flops_per_step=flops_per_step, next_step=nxt + 64 + timing_steps,
                    host_issue_ms=issue_ms, sum_timed_kernel_ms=(sum(per.values()) if per and mode != "frame" else None), graph_probe=probe)

    def roofline(self, m, full=True):
        timing, ms, flops_per_step = m["timing"], m["ms"], m["flops_per_step"]
        if not timing.get("dominant"):
            return None
        d = timing["dominant"]
        pm = pmc_info(d["kernel"])
        same_as_pmc = self.mode == "train" and self.config_id == 2 and not self.rays_override
        tr = (pm["hbm_bytes"], pm["source"]) if pm else None
        gbps, ghz, withheld = pmc_rates(pm, d["avg_launch_ms"]) if pm and same_as_pmc else (None, None, None)
        e2e = flops_per_step / (ms * 1e-3) / 1e12
        e2e_nom = timing.get("nominal_flops_per_step", 0.0) * (flops_per_step / max(timing.get("flops_per_step", 0.0), 1e-30)) / (ms * 1e-3) / 1e12
        x3 = "_x3" in d["kernel"]
        # a split-precision kernel issues SIX bf16 MACs per fp32-equivalent MAC: price it against the bf16 matrix peak
        ach, peak = (d["tflops"] * 6.0, PEAK_BF16_MFMA) if x3 else (d["tflops"], PEAK_F32_MFMA)
        if not full:
            return dict(bound="mfma", kernel=d["kernel"], achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                        avg_launch_ms=round(d["avg_launch_ms"], 4), launches=d["launches"],
                        end_to_end=dict(achieved=round(e2e, 2), frac=round(e2e / PEAK_F32_MFMA, 4), unit="TFLOP/s (executed, fp32-equivalent) of the fp32 MFMA peak"),
                        pipes=timing.get("pipes") if self.split else None)
        return dict(bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak,
                    traffic=tr[0] if tr else None,
                    traffic_unit="HBM bytes per SINGLE launch of the symbol (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE; launch-weighted mean over "
                                 "its instantiations / grid sizes: traffic_rows)",
                    traffic_source=tr[1] if tr else None, traffic_rows=pm["rows"] if pm else None, kernel=d["kernel"],
                    work="EXECUTED MACs (the MACs the kernel issues; kernel_macs in bench.py) x 2 x points per launch",
                    nominal=dict(achieved=d["nominal_tflops"] * (6.0 if x3 else 1.0), frac=d["nominal_tflops"] * (6.0 if x3 else 1.0) / peak,
                                 flops_per_launch=d["nominal_flops_per_launch"],
                                 note="SURVEY 8d per-point figures (D, S, C per pass) incl. the last-layer rows this kernel never issues"),
                    # counters of the same symbol from the committed PMC passes (profiles/): MFMA-busy share of the SIMD cycles and
                    # the HBM rate its traffic means at the launch duration measured here
                    # (rates only where the PMC passes profiled this very workload: the headline configuration)
                    mfma_util=pm["mfma_util"] if pm else None,
                    hbm_gbps=gbps, clock_ghz_from_pmc_cycles=ghz, pmc_rates_note=withheld,
                    kernel_choice=("the kernel SYMBOL with the largest total time per step (all its launch sizes together)"
                                   + (" among the kernels on the bf16 matrix pipes (split-precision mode; roofline.pipes has both pipes)" if self.split else "")),
                    avg_launch_ms=d["avg_launch_ms"], launches=d["launches"], flops_per_launch=d["flops_per_launch"],
                    fp32_equivalent_tflops=d["tflops"], share_of_timed_kernel_time=d["share_of_timed_kernel_time"],
                    pipes=timing.get("pipes"),
                    end_to_end=dict(achieved=e2e, frac=e2e / PEAK_F32_MFMA, unit="TFLOP/s", flops_per_step=flops_per_step,
                                    nominal_achieved=e2e_nom, nominal_frac=e2e_nom / PEAK_F32_MFMA,
                                    note="executed fp32-equivalent GEMM FLOPs of one step (2 x MACs x points of every timed launch) / "
                                         "ms_per_step, against the fp32 MFMA peak (a split-precision run can exceed it: its GEMMs run on "
                                         "the bf16 pipes; see roofline.pipes for the per-pipe rates)"),
                    peak_note=("bf16 MFMA dense peak (v_mfma_f32_32x32x16_bf16); achieved = 6 bf16 partial products per fp32-equivalent MAC"
                               if x3 else "fp32 MFMA dense peak (v_mfma_f32_32x32x2_f32)"))

    def describe(self):
        what = {"train": "full train step: render + errorondepth + surface_neighbour_error + loss + backward + Adam",
                "forward": "renderer forward only",
                "frame": "one 640x512 frame per step, forward only, hipGraph-captured %d-ray chunks" % self.chunk + ("" if self.frame_graph else " (eager)")}[self.mode]
        c = self.cfg
        per_gpu = self.n_rays if self.mode != "frame" else -(-self.H // self.ctx.world) * self.W
        return "BASELINE config %d: %s nets, %d rays x (%d+%d) samples per GPU, %s" % (self.config_id, c["name"], per_gpu, c["n_samples"], c["n_importance"], what)

    def metric(self):
        n, S = (self.n_rays, self.S)
        return {"train": "training rays/sec (%d rays x %d samples)" % (n, S), "forward": "forward rays/sec (%d rays x %d samples)" % (n, S),
                "frame": "full-frame render rays/sec (640x512, %d samples, %d-ray chunks)" % (S, self.chunk)}[self.mode]

    def close(self):
        import gc
        import torch
        self.renderer = self.trainer = self.eng = self.batches = self.frame_rays = None
        gc.collect()
        torch.cuda.empty_cache()


class ReferenceLoop:
    """The reference trainer's OWN training step against the drop-in renderer -- what a maintainer gets from INTEGRATION.md's import swap
    and nothing else: ``init_optimizer`` (trainer_endosurf.py:60-72: ``torch.optim.Adam(params=grad_vars, lr=lr_init)`` over the 82
    parameter tensors, torch's defaults), ``train_step`` (:94-104: zero_grad -> compute_loss -> backward -> step -> ``loss.item()``),
    ``compute_loss`` (:106-181: ``renderer(rays)``, torch loss arithmetic with ``F.l1_loss``, ``errorondepth``,
    ``surface_neighbour_error`` as three separate calls) and ``update_learning_rate`` (:183-203).  Nothing of endosurf_amd.trainer is used
    except the synthetic batch generator (the stand-in for ``Dataset.get_train_batch_data_by_index``).
    ``logging``: also the reference's per-step host traffic: ``cal_psnr`` on numpy copies (src/trainer/utils.py:340-353, three D2H
    copies) and the ``writer.add_scalar(tensor)`` calls of :165-179 (each one a D2H copy of a 0-d tensor = a host sync)."""

    W = dict(color=1.0, depth=1.0, sdf=1.0, angle=0.1, eikonal=0.1, surf_neig=0.1)      # base_pull.yml:23-29

    def __init__(self, ctx, logging=False, config_id=2, flat_adam=False, defer_eod=True):
        import torch
        from endosurf_amd import EndoSurfRenderer
        from endosurf_amd.trainer import SyntheticScene
        cfg = CONFIGS[config_id]
        torch.manual_seed(0)
        self.cfg, self.logging = cfg, bool(logging)
        self.renderer = EndoSurfRenderer(dict(render_cfg(cfg), defer_errorondepth=bool(defer_eod)), dict(NET_CFG, use_deform=cfg["use_deform"]),
                                         device=ctx.dev)
        train_params = self.renderer.get_train_params()
        grad_vars = []
        for key in train_params.keys():
            grad_vars += train_params[key]
        self.optimizer = torch.optim.Adam(params=grad_vars, lr=5e-4)
        if flat_adam:          # the ONE further line INTEGRATION.md offers a maintainer: the same update as one launch over the flat buffer
            from endosurf_amd.trainer import FlatAdam
            self.optimizer = FlatAdam(self.renderer, lr=5e-4)
        self.lr_init, self.n_iter = 5e-4, 100000
        scene = SyntheticScene(ctx.dev, seed=1234 + ctx.rank)
        self.batches = [scene.batch(cfg["rays"]) for _ in range(4)]
        self.t_issue = []          # per step: seconds from the step's first call to just before loss.item()
        self.scalars = 0

    def _log(self, t):
        """``SummaryWriter.add_scalar(tag, tensor)``: the value comes to the host (make_np: ``.detach().cpu().numpy()``)."""
        self.scalars += 1
        return t.detach().cpu().numpy()

    def compute_loss(self, data, global_step):
        import numpy as np
        import torch
        import torch.nn.functional as F
        r, w = self.renderer, self.W
        rays, color_gt, depth_gt, mask_gt, color_mask_gt = data["rays"], data["color"], data["depth"], data["mask"], data["color_mask"]
        ret = r(rays, iter_step=global_step)
        color_pred = ret["color_map"]
        color_error = (color_pred - color_gt) * color_mask_gt
        color_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction="sum") / (color_mask_gt.sum() + 1e-10)
        if self.logging:          # cal_psnr -> tensor2array right here, as in the reference (:137): the host waits for the render's forward
            a, b, m = (t.detach().cpu().numpy() for t in (color_pred, color_gt, color_mask_gt))
            _psnr = 20.0 * np.log10(1.0 / (((a - b) ** 2 * m).sum() / ((np.sum(m) + 1e-10) * 3.0)) ** 0.5)
        sdf_loss, angle_loss, valid_depth_region = r.errorondepth(rays, d_gt=depth_gt, mask=mask_gt, iter_step=global_step)
        depth_pred = ret["depth_map"]
        depth_error = (depth_pred - depth_gt) * valid_depth_region * mask_gt
        depth_loss = F.l1_loss(depth_error, torch.zeros_like(depth_error), reduction="sum") / ((valid_depth_region * mask_gt).sum() + 1e-10)
        eikonal_loss = ret["gradient_o_error"]
        surf_neig_loss = r.surface_neighbour_error(rays=rays, mask=mask_gt, iter_step=global_step, neighbour_rad=0.1)
        loss = (color_loss * w["color"] + depth_loss * w["depth"] + sdf_loss * w["sdf"] + angle_loss * w["angle"]
                + eikonal_loss * w["eikonal"] + w["surf_neig"] * surf_neig_loss)
        if self.logging:          # (the add_scalar reads sit BEFORE the backward, as in the reference: the host waits for the forward here)
            for t in (color_loss, sdf_loss, angle_loss, depth_loss, eikonal_loss, surf_neig_loss, loss, ret["s_val"].mean(),
                      (ret["cdf"][:, :1] * mask_gt).sum() / (mask_gt.sum() + 1e-10), (ret["weight_max"] * mask_gt).sum() / (mask_gt.sum() + 1e-10)):
                self._log(t)
        return loss

    def update_learning_rate(self, global_step, warm_up_end=5000, alpha=0.05):
        import numpy as np
        if global_step < warm_up_end:
            f = global_step / warm_up_end
        else:
            f = (np.cos(np.pi * (global_step - warm_up_end) / (self.n_iter - warm_up_end)) + 1.0) * 0.5 * (1 - alpha) + alpha
        for g in self.optimizer.param_groups:
            g["lr"] = self.lr_init * f

    def train_step(self, i):
        """One iteration of the reference's main loop (trainer_basic.py:86-105): train_step + update_learning_rate; -> loss (host float)."""
        data = self.batches[i % len(self.batches)]
        t0 = time.perf_counter()
        self.optimizer.zero_grad()
        loss = self.compute_loss(data, i + 1)
        loss.backward()
        self.optimizer.step()
        self.t_issue.append(time.perf_counter() - t0)
        v = loss.item()
        self.update_learning_rate(i + 1)
        return v

    def measure(self, warmup, steps, timing_steps=3):
        """``ms_per_step``: the data-independent step (all 128 marching proposals of every ray, like the headline's ``value``); the package
        default (early exit at each ray's first sign change, bit-identical results) is timed next to it as ``with_early_exit``."""
        import torch
        from endosurf_amd import _lib
        eng = self.renderer.engine
        march_block = eng.march_block
        eng.march_block = 0
        for i in range(warmup):
            self.train_step(i)
        torch.cuda.synchronize()
        self.t_issue, self.scalars = [], 0
        calls0 = _lib.calls
        t0 = time.perf_counter()
        for i in range(steps):
            self.train_step(warmup + i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        calls = (_lib.calls - calls0) / steps
        # (a logging step blocks on its host reads in the middle: "time to enqueue" is only defined for the plain step)
        issue = None if self.logging else sorted(self.t_issue)[len(self.t_issue) // 2] * 1e3
        syncs = 1 + (self.scalars // steps + 3 if self.logging else 0)
        nxt = warmup + steps
        timing = kernel_timing(eng, self.train_step, nxt, timing_steps, self.cfg["use_deform"])
        nxt += timing_steps
        per = timing.get("per_step_ms") or {}
        early = None
        if march_block:
            eng.march_block = march_block
            self.train_step(nxt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                self.train_step(nxt + 1 + i)
            torch.cuda.synchronize()
            dte = time.perf_counter() - t0
            early = dict(ms_per_step=dte / steps * 1e3, value=self.cfg["rays"] * steps / dte, block=march_block,
                         note="the package default; bit-identical results, data-dependent saving (how many rays have passed their first sign "
                              "change after each block of proposals)")
        opt = self.optimizer
        tail = (self.renderer.__dict__.get("_live_tail") or (None,))[0]
        return dict(ms_per_step=dt / steps * 1e3, value=self.cfg["rays"] * steps / dt, unit="rays/s", steps=steps, warmup=warmup,
                    ray_marching="all 128 proposals of every ray (data independent, as the reference)", with_early_exit=early,
                    host_issue_ms=issue, host_syncs_per_step=syncs, library_calls_per_step=calls,
                    sum_timed_kernel_ms=sum(per.values()) if per else None, kernel_ms_per_step=per or None,
                    aux_rows_in_render_workspace=(tail.cap if tail is not None else 0),
                    optimizer=("endosurf_amd.trainer.FlatAdam(renderer, lr): the same update as one launch" if "params" not in opt.param_groups[0] else
                               "torch.optim.Adam(params=grad_vars, lr) over %d tensors, torch defaults (%s)" % (
                                   len(opt.param_groups[0]["params"]),
                                   "foreach" if opt.param_groups[0].get("foreach") in (None, True) and not opt.param_groups[0].get("fused") else "fused")))


def refseq_launches_from_profile():
    """Kernel launches per step of the reference loop from the newest committed rocprofv3 kernel statistics of ``bench.py --refseq-only
    plain --steps 10 --warmup 3`` (tools/profile_round.sh: 3 warm-up + 10 timed + 3 instrumented + 1 + 10 early-exit steps = 27 steps):
    all kernels, the library's (es::), the framework's element-wise / reduction kernels (at::native, of which the optimiser's multi-tensor
    ones).  None if no profile is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_plain_kernel_stats.csv")), reverse=True)
    if not files:
        return None
    rows = list(csv.DictReader(open(files[0])))
    n = 27.0
    cnt = lambda pred: round(sum(int(r["Calls"]) for r in rows if pred(r["Name"])) / n, 1)
    ms = lambda pred: round(sum(float(r["TotalDurationNs"]) for r in rows if pred(r["Name"])) / 1e6 / n, 3)
    return dict(source=os.path.basename(files[0]), steps_profiled=int(n), all=cnt(lambda s: True), library=cnt(lambda s: "es::" in s),
                framework=cnt(lambda s: "at::native" in s), of_which_optimizer=cnt(lambda s: "multi_tensor" in s),
                kernel_ms_per_step=dict(all=ms(lambda s: True), library=ms(lambda s: "es::" in s), framework=ms(lambda s: "at::native" in s),
                                        of_which_optimizer=ms(lambda s: "multi_tensor" in s)))


def reference_call_sequence(ctx, args, steps=None):
    """extras.reference_call_sequence: the reference trainer's own loop through the drop-in, config 2, same synthetic
    batches as the headline -- without and with the reference's per-step logging traffic."""
    import gc
    import torch
    short = args.steps < 10
    steps = steps or (4 if short else 20)
    out = {}
    for name, logging, flat in (("plain", False, False), ("with_reference_logging", True, False), ("plain_with_flat_adam", False, True)):
        loop = ReferenceLoop(ctx, logging=logging, flat_adam=flat)
        out[name] = loop.measure(3, steps, timing_steps=2 if short else 3)
        loop = None
        gc.collect()
        torch.cuda.empty_cache()
    out["launches_per_step"] = refseq_launches_from_profile()
    out["what"] = ("the reference trainer's own step (trainer_endosurf.py:60-72, 94-104, 106-181, 183-203) with "
                   "src.renderer.endosurf.EndoSurfRenderer replaced by endosurf_amd.EndoSurfRenderer and nothing else: renderer(rays) -> "
                   "errorondepth -> surface_neighbour_error as three calls, torch loss arithmetic, loss.backward(), torch.optim.Adam.step() "
                   "over the parameter tensors, loss.item(); 'with_reference_logging' adds cal_psnr's three D2H copies and the ten "
                   "add_scalar(tensor) host syncs of :165-179; 'plain_with_flat_adam' replaces the optimiser line by endosurf_amd.trainer.FlatAdam (the one "
                   "further change INTEGRATION.md offers).  ms_per_step is wall time per iteration (every iteration ends in a host "
                   "sync, so host issue and GPU work of consecutive steps do not overlap)")
    return out


def collective_proof(ctx, wl, dt_local):
    """What the collective itself saw (N > 1, or the forced one-rank RCCL group): an all-reduce of ones, every rank's own ms per step,
    and the time of the step's ONE data-path collective (the all-reduce of the flat gradient bucket) from HIP events around it."""
    import torch
    import torch.distributed as dist
    from endosurf_amd.parallel import allreduce_flat
    ones = torch.ones(1, device=ctx.dev)
    dist.all_reduce(ones)
    per_rank = ctx.gather_over_ranks(dt_local)
    flat = torch.zeros(wl.eng.n_param, device=ctx.dev)
    for _ in range(3):
        allreduce_flat(flat, force=True)
    ctx.barrier()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    a.record()
    for _ in range(n):
        allreduce_flat(flat, force=True)
    b.record()
    torch.cuda.synchronize()
    ar_ms = ctx.max_over_ranks(a.elapsed_time(b) / n)
    # the same collective INSIDE training steps: HIP events around it on the launch stream.  In a step it is issued behind the last launch
    # of the backward (es_weightnorm_backward writes the bucket last) and the Adam launch consumes its result, so all of it is exposed;
    # what the events add to the back-to-back figure is the skew between the ranks at that point of the step.
    in_step = None
    if wl.mode == "train" and not wl.use_graph:
        wl.trainer.allreduce_events = []
        for i in range(4):
            wl.step(10_000 + i)
        torch.cuda.synchronize()
        ts = [x.elapsed_time(y) for x, y in wl.trainer.allreduce_events]
        wl.trainer.allreduce_events = None
        if ts:
            in_step = ctx.max_over_ranks(sorted(ts)[len(ts) // 2])
    # the OTHER all-reduce mode on a few steps (pipelined buckets if the run used one bucket, and vice versa): the comparison a cold
    # N > 1 run needs in order to decide the default, at the price of ~10 steps
    other = None
    if wl.mode == "train" and not wl.use_graph and dist.get_world_size() > 1:
        tr = wl.trainer
        tr.overlap_allreduce = not tr.overlap_allreduce
        try:
            for i in range(3):
                wl.step(20_000 + i)
            dto, _ = wl.timed(20_003, 8)
            other = dict(overlap_allreduce=tr.overlap_allreduce, ms_per_step=dto / 8 * 1e3, steps=8,
                         pipelined_steps=int(tr.pipelined_steps))
        finally:
            tr.overlap_allreduce = not tr.overlap_allreduce
    # the replicas after all the steps above: every rank's flat parameter buffer against rank 0's, bit for bit (the summed bucket and the
    # update are the same on every rank, so any difference is a bug)
    mine = wl.renderer.model._flat.detach()
    ref = mine.clone()
    dist.broadcast(ref, src=0)
    diff = (mine != ref).sum().to(torch.float64).reshape(1)
    dist.all_reduce(diff)
    return dict(ranks_seen_by_collective=int(round(float(ones.item()))), bucket_bytes=4 * wl.eng.n_param,
                replicas_bit_identical=bool(float(diff.item()) == 0.0), replica_words_differing=int(diff.item()),
                allreduce_ms=ar_ms, allreduce_other_mode=other, allreduce_in_step_ms=in_step, allreduce_exposed_ms=in_step,
                allreduce_hidden_ms=(None if in_step is None else (max(0.0, ar_ms - in_step) if wl.trainer.overlap_allreduce else 0.0)),
                allreduce_overlap_note="exposed = the collective as the step's launch stream sees it (median of 4 steps, MAX over ranks): the bucket is "
                "complete only behind the last backward launch and Adam needs all of it, so nothing of a ONE-bucket step can hide it (hidden = 0); "
                "--overlap-allreduce issues two of three buckets on a side stream under the remaining weight-gradient launches (DESIGN 6): "
                "allreduce_other_mode times a few steps of the mode this run did not use",
                allreduce_note="mean of %d back-to-back all-reduces of the 6.6 MB flat gradient bucket, HIP events on the "
                "launch stream, MAX over ranks (in a step it is issued once, after the last weight-gradient launch)" % n,
                per_rank_seconds=per_rank)


def run_extras(ctx, args, partial):
    """Everything else the builder reports, in the SAME driver-run command: cfg2 forward-only, cfg2 in the
    opt-in split-precision mode, cfg3, cfg4 and one cfg5 frame -- each a fresh renderer, a few steps, OUTSIDE the headline's timed region.
    N > 1: only the configurations BASELINE.json names for several GPUs (cfg4: data-parallel training, cfg5: the frame's rows split
    over the ranks + one all-gather).  ``partial`` is filled as the extras finish (the watchdog prints what is there)."""
    import torch
    short = args.steps < 10          # (tests run the command with a handful of steps)
    plan = [("forward", dict(config_id=2, mode="forward"), 10, 5 if short else 20, 3),
            ("split_precision_train", dict(config_id=2, mode="train", split=True), 3, 5 if short else 20, 3),
            ("cfg3", dict(config_id=3, mode="train"), 2, 3 if short else 10, 2),
            ("cfg4", dict(config_id=4, mode="train"), 3, 5 if short else 20, 3),
            ("cfg5_frame", dict(config_id=5, mode="frame", chunk=args.chunk), 0, 1 if short else 2, 0)]
    if ctx.world > 1:
        plan = [p for p in plan if p[0] in ("cfg4", "cfg5_frame")]
        partial["skipped_at_n_gt_1"] = "forward, split_precision_train, cfg3: single-GPU lines (reported by the N = 1 run)"
    ok = True
    if ctx.world == 1:
        try:
            partial["reference_call_sequence"] = reference_call_sequence(ctx, args)
        except Exception as e:
            partial["reference_call_sequence"] = dict(error="%s: %s" % (type(e).__name__, str(e)[:500]))
    for name, kw, warm, steps, tsteps in plan:
        if ctx.dist_on:      # a rank that failed an extra must not leave the others waiting inside the next one's collectives
            flag = torch.tensor([1.0 if ok else 0.0], device=ctx.dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            if float(flag.item()) < 1.0:
                partial[name] = dict(error="skipped: an earlier extra failed on some rank")
                continue
        t0 = time.perf_counter()
        wl = None
        try:
            wl = Workload(ctx, **kw)
            if wl.mode == "frame":      # warm-up = a few chunks (lazy init + the graph capture), then ONE timed frame
                wl.renderer.render_frames(wl.frame_rays.reshape(-1, 9)[:2 * args.chunk], iter_step=1, ray_chunk=args.chunk, perturb_overwrite=False)
            m = wl.measure(warm, steps, timing_steps=tsteps, early_exit_extra=False)
            if ctx.rank == 0:
                partial[name] = dict(ms_per_step=m["ms"], value=m["value"], unit="rays/s", steps=steps, warmup=warm, metric=wl.metric(),
                                     workload=wl.describe(), n_gpus=ctx.world, roofline=wl.roofline(m, full=False),
                                     kernel_ms_per_step=m["timing"].get("per_step_ms"), host_issue_ms=m["host_issue_ms"],
                                     sum_timed_kernel_ms=m["sum_timed_kernel_ms"],
                                     captured=(bool((getattr(wl.renderer, "_fwd_graph", None) or {}).get("graph")) if wl.mode == "forward" else None),
                                     seconds=None)
        except Exception as e:      # an extra must never cost the headline line
            ok = False
            partial[name] = dict(error="%s: %s" % (type(e).__name__, str(e)[:500]))
        finally:
            if wl is not None:
                wl.close()
            if ctx.rank == 0 and isinstance(partial.get(name), dict):
                partial[name]["seconds"] = round(time.perf_counter() - t0, 2)
    return partial


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configuration (2 = the metric's)")
    ap.add_argument("--rays", type=int, default=None, help="override the configuration's rays per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default=None, choices=["train", "forward", "frame"])
    ap.add_argument("--refseq-only", choices=["plain", "logging", "flat_adam"], default=None,
                    help="profiling aid: run ONLY the reference trainer's own loop through the drop-in (ReferenceLoop) and print its record")
    ap.add_argument("--no-graph", action="store_true", help="frame mode: eager launches instead of the captured hipGraph")
    ap.add_argument("--schedule", default="fused", choices=["fused", "plain"])
    ap.add_argument("--no-defer-eod", action="store_true", help="--refseq-only: render_cfg['defer_errorondepth'] = False (A/B of the deferred errorondepth evaluation)")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the extra early-exit timing and the extras (profiling runs: every launch of the process then belongs to the "
                         "headline workload)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra lines (forward / split precision / cfg3 / cfg4 / cfg5 frame)")
    ap.add_argument("--extras-timeout", type=float, default=420.0, help="watchdog: seconds the extras may take before the line is printed without them")
    ap.add_argument("--split-precision", action="store_true",
                    help="OPT-IN extra line, never the headline: large no-grad SDF queries and the weight-gradient GEMMs on the bf16 matrix "
                         "pipes with exact 3-way operand splitting (csrc/query_x3.hip, wgrad.hip); everything else stays fp32 MFMA")
    ap.add_argument("--graph", action="store_true",
                    help="train mode: the whole training step captured once in a hipGraph and replayed (Trainer.train_step_graph); the "
                         "per-kernel timers then run on a few eager steps after the timed region")
    ap.add_argument("--graph-probe", action="store_true",
                    help="also time the step replayed from the whole-step hipGraph and apply the host-bound fallback rule (always on at N > 1)")
    ap.add_argument("--no-pin", action="store_true", help="N > 1: do not pin the ranks to disjoint host-core sets")
    ap.add_argument("--overlap-allreduce", action="store_true",
                    help="data-parallel training: the gradient all-reduce as a pipeline of three buckets, two of them on a side stream under "
                         "the remaining weight-gradient launches (Trainer(overlap_allreduce=True); opt-in until a multi-GPU run has measured it)")
    ap.add_argument("--chunk", type=int, default=2048, help="frame mode: rays per chunk (the reference's demo.ray_batch is 2048)")
    args = ap.parse_args()

    backend = os.environ.get("ES_DIST_BACKEND", "nccl")
    if args.gpus > 1 and backend == "nccl":
        # one rank per GPU over RCCL: fail before any rendezvous, with the numbers in the message (never fold ranks onto one device)
        import torch
        if torch.cuda.device_count() < args.gpus:
            sys.exit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node "
                     f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')!r}); RCCL needs one GPU per rank")
    if args.gpus > 1 and "RANK" not in os.environ:
        # launched directly: one rank per GPU under torch.distributed.run on this node (RCCL; rendezvous on 127.0.0.1)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    from endosurf_amd import parallel
    # one rank per GPU over RCCL ("nccl" on ROCm); ES_DIST_BACKEND=gloo lets the tests drive the N > 1 path on a single GPU
    # ES_FORCE_DIST=1 under a one-rank torch.distributed.run: the N-rank code path (RCCL rendezvous, broadcast, barrier, the gradient
    # all-reduce, MAX over ranks) on a single GPU -- the smoke test of the scaling runs (tests/test_gpu_bench_dp.py)
    force_dist = args.gpus == 1 and os.environ.get("ES_FORCE_DIST") == "1" and "RANK" in os.environ
    rank, world, local = parallel.init_distributed(backend if (args.gpus > 1 or force_dist) else None, force=force_dist)
    dist_on = world > 1 or force_dist
    assert world == max(1, args.gpus), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # ES_DIST_BACKEND=gloo is the test mode in which several ranks may share one GPU (tests/test_gpu_bench_dp.py)
    local = parallel.local_device(local, world) if (world == 1 or backend == "nccl") else local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ctx = Ctx(dev, rank, world, dist_on, force_dist, backend, args.schedule, overlap_allreduce=args.overlap_allreduce)
    # cold-run kit: every local rank on its own host cores (within the mask this process was given)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    affinity = (parallel.pin_rank_to_cores(int(os.environ.get("LOCAL_RANK", "0")), local_world) if world > 1 and not args.no_pin
                else dict(pinned=False, reason="one rank" if world == 1 else "--no-pin"))

    if args.refseq_only:
        loop = ReferenceLoop(ctx, logging=args.refseq_only == "logging", flat_adam=args.refseq_only == "flat_adam", defer_eod=not args.no_defer_eod,
                             config_id=args.config if args.config in (2, 3, 4) else 2)
        print(json.dumps(loop.measure(args.warmup, args.steps)), flush=True)
        return

    # ---- headline ---------------------------------------------------------------------------------------------------------------
    wl = Workload(ctx, args.config, mode=args.mode, split=args.split_precision, rays=args.rays, chunk=args.chunk, graph=args.graph,
                  frame_graph=not args.no_graph)
    wl.graph_probe = (world > 1 or args.graph_probe) and wl.mode == "train" and not args.graph and not args.split_precision
    m = wl.measure(args.warmup, args.steps, early_exit_extra=not args.headline_only)
    proof = collective_proof(ctx, wl, m["dt_local"]) if dist_on else None
    probe = m.get("graph_probe")
    # per-rank diagnostics of a cold N > 1 run, gathered once (every rank contributes; rank 0 prints)
    per_rank = ctx.gather_obj(dict(rank=rank, host_issue_ms=m["host_issue_ms"], affinity=affinity,
                                   gpu_idle_ms=(probe or {}).get("gpu_idle_ms"), ms_per_step=m["dt_local"] / args.steps * 1e3))
    cfg, mode, timing = wl.cfg, wl.mode, m["timing"]
    out = None
    if rank == 0:
        per_rank_ms = [s / args.steps * 1e3 for s in proof["per_rank_seconds"]] if proof else [m["dt_local"] / args.steps * 1e3]
        out = dict(metric=wl.metric(), value=m["value"], unit="rays/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=m["ms"],
                   higher_is_better=True, scaling="strong" if mode == "frame" else "weak", vs_baseline=None,
                   dtype="f32" if not args.split_precision else "f32 (OPT-IN split precision: large SDF queries, the point-evaluation chains and the "
                                                                "weight-gradient GEMMs as 3 x bf16 planes, 6 partial products, fp32 accumulate; not the "
                                                                "headline configuration)",
                   data="synthetic",
                   config=dict(workload=wl.describe(),
                               baseline_config=args.config, use_deform=cfg["use_deform"], split_precision=bool(args.split_precision),
                               whole_step_hipgraph=(bool(args.graph) or bool((probe or {}).get("use_graph"))) and mode == "train",
                               graph_fallback=((probe or {}).get("rule") if (probe or {}).get("use_graph") else None),
                               ray_marching="all 128 proposals of every ray (data independent, as the reference)" if mode == "train" else None,
                               with_early_exit=m["with_early_exit"], rays_per_gpu=wl.n_rays if mode != "frame" else None,
                               rays_per_step_whole_job=wl.rays_per_step_job,
                               frame_rows_per_rank=([parallel.frame_rows(wl.H, r, world)[1] for r in range(world)] if mode == "frame" else None),
                               collective=("rccl all-reduce forced at world 1" if force_dist else (
                                   ("%s all-reduce of the flat 6.6 MB gradient bucket per step" % backend) if world > 1 and mode == "train" else (
                                       "%s all-gather of the row slabs per frame" % backend if world > 1 and mode == "frame" else None))),
                               samples_per_ray=wl.S, parallelism=f"dp{world}", overlap_allreduce=bool(args.overlap_allreduce),
                               weights="reference init, torch.manual_seed(0)", algorithmic_gflop_per_ray=algorithmic_gflop_per_ray(cfg)),
                   ms_per_step_min=min(per_rank_ms), ms_per_step_max=max(per_rank_ms),
                   ranks_seen_by_collective=proof["ranks_seen_by_collective"] if proof else None,
                   allreduce_ms=proof["allreduce_ms"] if proof else None,
                   collective_proof=({k: v for k, v in proof.items() if k != "per_rank_seconds"} if proof else None),
                   host_issue_ms=m["host_issue_ms"], sum_timed_kernel_ms=m["sum_timed_kernel_ms"],
                   host_issue_ms_min=min(r["host_issue_ms"] for r in per_rank) if per_rank[0]["host_issue_ms"] is not None else None,
                   host_issue_ms_max=max(r["host_issue_ms"] for r in per_rank) if per_rank[0]["host_issue_ms"] is not None else None,
                   gpu_idle_ms_min=min(r["gpu_idle_ms"] for r in per_rank) if per_rank[0]["gpu_idle_ms"] is not None else None,
                   gpu_idle_ms_max=max(r["gpu_idle_ms"] for r in per_rank) if per_rank[0]["gpu_idle_ms"] is not None else None,
                   gpu_idle_note="per rank: eager ms_per_step - ms_per_step of the same step replayed from the whole-step hipGraph (which needs no "
                                 "host work between its launches): the time per step the GPU waited for that rank's host",
                   graph_probe=({k: v for k, v in probe.items() if k not in ("dt", "dt_local", "gpu_idle_ms")} if probe else None),
                   per_rank=per_rank,
                   host_note="host_issue_ms: CPU time to enqueue one step into an empty queue (median of 5); sum_timed_kernel_ms: the MLP chain / "
                             "query / weight-gradient launches of one step (HIP events; the two front-end chains of a training step overlap)",
                   roofline=wl.roofline(m), kernel_ms_per_step=timing.get("per_step_ms"), kernel_symbols=timing.get("symbols"),
                   kernel_launch_groups=timing.get("launch_groups"),
                   hbm_accounting=(hbm_accounting(wl.n_rays * wl.S, 3 * wl.n_rays) if (args.config == 2 and mode == "train" and not args.rays
                                                                                       and not args.split_precision and args.schedule == "fused") else None),
                   hbm_accounting_note="algorithmic HBM bytes per launch (bench.py HBM_BYTES_PER_POINT: every saved stack written once / read once, "
                                       "per point, x the points of the launch) beside the rocprofv3 traffic of the same symbol from the committed PMC "
                                       "summary (FETCH_SIZE x 2 + WRITE_SIZE); ratio = measured / algorithmic",
                   cpu_baseline=None, extras=None)
    headline_is_default = args.config == 2 and mode == "train" and not args.split_precision and not args.rays
    want_extras = headline_is_default and not (args.no_extras or args.headline_only)
    wl.close()

    # ---- after the headline, outside every timed region: the CPU baseline (rank 0 at N = 1) and the extras -------------------------
    # A watchdog prints the line without the unfinished part if that section hangs (an RCCL collective with a missing rank would block for
    # minutes): the headline above is never at the mercy of an extra.
    import threading
    lock, state = threading.Lock(), dict(printed=False)
    extras = {}

    def emit(note=None):
        with lock:
            if state["printed"]:
                return
            state["printed"] = True
            if rank == 0:
                if want_extras:
                    out["extras"] = dict(extras, **({"watchdog": note} if note else {}))
                print(json.dumps(out), flush=True)

    def watchdog():
        emit("extras did not finish within %.0f s: printed without the unfinished ones" % args.extras_timeout)
        os._exit(0)

    timer = threading.Timer(args.extras_timeout if rank == 0 else args.extras_timeout + 10.0, watchdog)
    timer.daemon = True
    timer.start()
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline()
        # the extras follow ~30 s of 16-thread CPU work: let the host settle and keep torch's intra-op pool out of the launch thread's way
        torch.set_num_threads(1)
        time.sleep(1.0)
    if want_extras:
        run_extras(ctx, args, extras)
    timer.cancel()
    emit()
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
