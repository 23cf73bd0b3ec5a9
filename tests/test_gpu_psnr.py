"""PSNR parity (BASELINE.json "at matched PSNR"): the HIP renderer trained on the build-owned synthetic scene follows the PSNR curve
of the REFERENCE renderer trained on CPU (tools/psnr_reference.py) with the same initial weights, batches and random draws, through
the steep part and onto the PLATEAU of a complete schedule (1500 iterations, warm-up + cosine decay to 5 %).

What "matched" can mean is bounded by the reference itself: training is chaotic, and fp32 runs of the reference that differ only in
the intra-op thread count (another GEMM summation order; tests/golden/psnr_reference_long.npz = 4 threads, psnr_reference_t*.npz =
other counts) drift apart by up to 8 dB at single evaluation points of the steep phase and still end ~2 dB apart on the plateau.
The test therefore asserts
  (1) the same start and the same early trajectory (before rounding differences have been amplified);
  (2) the plateau (mean of the last evaluations) within max(0.5 dB, 2.5 sample standard deviations) of the mean plateau of the
      reference's own runs -- a new run of the REFERENCE lands outside the [min, max] of n earlier runs with probability 2 / (n + 1),
      so the band itself would be a flaky criterion even for the reference;
  (3) the same final loss level.
The HIP run uses the deterministic reduction mode, so it is itself bit-reproducible (tests/test_gpu_determinism.py)."""
import os

import numpy as np
import pytest
import torch

import synth_scene
from gpu_util import renderer_for

pytestmark = pytest.mark.gpu
GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = os.path.join(GOLD_DIR, "psnr_reference_long.npz")
N_TAIL = 4          # evaluations averaged on the plateau (iterations 1350..1500)
MARGIN_DB = 0.5


def _reference_runs():
    """Every committed run of the reference on this schedule: {name: curve}; the 4-thread run first."""
    import glob
    runs = {"psnr_reference_long": np.load(GOLD)["curve"]}
    for f in sorted(glob.glob(os.path.join(GOLD_DIR, "psnr_reference_t*.npz"))):       # (incomplete / differently scheduled runs are skipped below)
        c = np.load(f)["curve"]
        if len(c) == len(runs["psnr_reference_long"]):
            runs[os.path.basename(f)[:-4]] = c
    return runs


def _plateau_band(runs):
    """(lo, hi, plateau values): mean of the reference runs' plateaus +- max(0.5 dB, 2.5 x their sample standard deviation)."""
    ends = [float(np.mean(c[-N_TAIL:, 1])) for c in runs.values()]
    tol = max(MARGIN_DB, 2.5 * float(np.std(ends, ddof=1))) if len(ends) > 1 else MARGIN_DB
    return float(np.mean(ends)) - tol, float(np.mean(ends)) + tol, ends


def _train(n_iter, n_rays, weight_seed, sched_seed, eval_its, deterministic=True, split=False, lr=5e-4, reference_sequence=False, stop_at=None,
           keep=None):
    """``reference_sequence``: the reference trainer's own call pattern -- renderer(rays), errorondepth, surface_neighbour_error as three
    calls, torch loss arithmetic, torch.optim.Adam over the parameter tensors -- instead of the fused step."""
    from endosurf_amd.trainer import Trainer, cal_psnr
    r = renderer_for(weight_seed, "init", True)
    r.engine.deterministic = deterministic
    r.engine.split_precision = split
    if keep is not None:
        keep.append(r)
    tr = Trainer(r, lr=lr, n_iter=n_iter, warm_up_end=max(n_iter // 10, 1), lr_alpha=0.05, fused=not reference_sequence,
                 flat_adam=not reference_sequence)
    sched = synth_scene.schedule(sched_seed, n_iter, n_rays)
    ev = {k: torch.from_numpy(v).cuda() for k, v in synth_scene.eval_batch().items()}
    eval_its = set(int(i) for i in eval_its)
    curve, losses = [], []
    for it in range(1, (stop_at or n_iter) + 1):
        b = {k: torch.from_numpy(v).cuda() for k, v in sched[it - 1].items()}
        tr.update_learning_rate(it)
        loss, _, _ = tr.train_step(b, it, u_perturb=b["u_perturb"], u_neigh=b["u_neigh"])
        losses.append(loss)
        if it in eval_its:
            with torch.no_grad():
                e = r(ev["rays"], iter_step=it, perturb_overwrite=False)
            curve.append((it, float(cal_psnr(e["color_map"], ev["color"], ev["mask"]))))
    return np.array(curve), torch.stack(losses).cpu().numpy()


@pytest.mark.skipif(not os.path.exists(GOLD), reason="reference PSNR curve not generated (tools/psnr_reference.py)")
def test_psnr_curve_matches_reference_to_the_plateau():
    g = np.load(GOLD)
    n_iter, n_rays = int(g["n_iter"]), int(g["n_rays"])
    ref_curve, ref_loss = g["curve"], g["loss"]
    curve, losses = _train(n_iter, n_rays, int(g["weight_seed"]), int(g["sched_seed"]), ref_curve[:, 0])
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    runs = _reference_runs()
    lo, hi, ends = _plateau_band(runs)
    others = [c for k, c in runs.items() if k != "psnr_reference_long"]
    spread_max = max([float(np.max(np.abs(c[:, 1] - ref_curve[:, 1]))) for c in others], default=None)
    np.savez(os.path.join(out, "psnr_hip.npz"), curve=curve, loss=losses, ref_curve=ref_curve, ref_loss=ref_loss,
             **{f"ref_{k}": c for k, c in runs.items()})
    assert np.array_equal(curve[:, 0], ref_curve[:, 0])
    d = curve[:, 1] - ref_curve[:, 1]
    # (1) identical start (same weights) and the same early trajectory (iterations 1..60)
    assert abs(d[0]) < 0.02, d[0]
    early = ref_curve[:, 0] <= 60
    assert np.max(np.abs(d[early])) < 0.1, d[early]
    # (2) the plateau: within the reference's own run-to-run variability of its mean plateau
    end = float(np.mean(curve[-N_TAIL:, 1]))
    assert lo < end < hi, (end, lo, hi, ends)
    assert end > curve[0, 1] + 15.0, "training must reach the reference's quality level"
    # in between the curves may only differ as much as the reference differs from itself (+ margin)
    if spread_max is not None:
        assert np.max(np.abs(d)) < max(3.0, 1.5 * spread_max), (float(np.max(np.abs(d))), spread_max)
    # (3) losses: identical first step, same level at the end
    assert abs(losses[0] - ref_loss[0]) < 2e-3 * max(1.0, abs(ref_loss[0]))
    assert abs(np.mean(losses[-100:]) - np.mean(ref_loss[-100:])) < 0.1 * abs(np.mean(ref_loss[-100:])) + 0.005


@pytest.mark.skipif(not os.path.exists(GOLD), reason="reference PSNR curve not generated (tools/psnr_reference.py)")
def test_psnr_early_curve_through_the_reference_call_sequence():
    """The reference trainer's OWN call sequence through the drop-in (three renderer calls, torch.optim.Adam; from the second step on the
    auxiliary calls' points ride in the render workspace's tail) follows the reference's PSNR curve through the first 150 iterations of
    the schedule as the fused step does: identical start, < 0.1 dB up to iteration 60, and the same losses as the fused step's."""
    g = np.load(GOLD)
    n_iter, n_rays, ref_curve = int(g["n_iter"]), int(g["n_rays"]), g["curve"]
    its = ref_curve[ref_curve[:, 0] <= 150, 0]
    keep = []
    curve, losses = _train(n_iter, n_rays, int(g["weight_seed"]), int(g["sched_seed"]), its, reference_sequence=True, stop_at=150, keep=keep)
    assert keep[0].tails_made >= 148                    # every step but the first went through the render workspace's tail
    d = curve[:, 1] - ref_curve[:len(curve), 1]
    assert abs(d[0]) < 0.02 and np.max(np.abs(d[its <= 60])) < 0.1, d
    fused_curve, fused_losses = _train(n_iter, n_rays, int(g["weight_seed"]), int(g["sched_seed"]), its, stop_at=150)
    assert abs(losses[0] - fused_losses[0]) < 1e-5 * max(1.0, abs(fused_losses[0]))
    assert np.max(np.abs(losses[:20] - fused_losses[:20])) < 2e-3 * np.max(np.abs(fused_losses[:20]))
    assert np.max(np.abs(curve[:, 1] - fused_curve[:, 1])[its <= 60]) < 0.1


def test_psnr_run_is_reproducible_in_deterministic_mode():
    """Two deterministic runs of the first 60 iterations give the same losses and PSNR values bit for bit."""
    a = _train(60, 256, 7, 11, [1, 30, 60])
    b = _train(60, 256, 7, 11, [1, 30, 60])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def _distribution(n_runs, split, tag):
    """n_runs HIP trainings in the BENCHMARKED reduction mode (fp32 atomics) against every committed run of the reference: the
    assertions of the distributional form of "matched PSNR" (docstring of the fp32 test below); -> the statistics (also written to
    gpurun_out/psnr_stats{tag}.json)."""
    g = np.load(GOLD)
    n_iter, n_rays, ref_curve = int(g["n_iter"]), int(g["n_rays"]), g["curve"]
    runs = _reference_runs()
    ref_pl = np.array([float(np.mean(c[-N_TAIL:, 1])) for c in runs.values()])
    hip_pl, curves = [], []
    for _ in range(n_runs):
        curve, _ = _train(n_iter, n_rays, int(g["weight_seed"]), int(g["sched_seed"]), ref_curve[:, 0], deterministic=False, split=split)
        d = curve[:, 1] - ref_curve[:, 1]
        assert abs(d[0]) < 0.02 and np.max(np.abs(d[ref_curve[:, 0] <= 60])) < 0.1, d[:8]
        hip_pl.append(float(np.mean(curve[-N_TAIL:, 1])))
        curves.append(curve)
    hip_pl = np.array(hip_pl)
    # standard error of the difference of the two means.  The HIP runs' variance is estimated from 3-5 runs: when they happen to land close
    # together the sample variance understates the run-to-run scatter of a chaotic training (the reference's six runs spread 1.6 dB), and
    # the test would flag a correct implementation -- so the reference's variance is the floor of the HIP runs' estimate
    hip_var = max(float(hip_pl.var(ddof=1)), float(ref_pl.var(ddof=1)))
    se = float(np.sqrt(hip_var / len(hip_pl) + ref_pl.var(ddof=1) / len(ref_pl)))
    delta = float(hip_pl.mean() - ref_pl.mean())
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    import json
    stats = dict(reference_runs=list(runs), reference_plateaus=ref_pl.tolist(), hip_plateaus=hip_pl.tolist(), delta_mean_db=delta,
                 standard_error_db=se, hip_std_db=float(hip_pl.std(ddof=1)), reference_std_db=float(ref_pl.std(ddof=1)),
                 mode=("split precision (bf16 x 3 chains, queries and weight gradients)" if split else "fp32") + ", atomic reductions")
    with open(os.path.join(out, f"psnr_stats{tag}.json"), "w") as f:
        json.dump(stats, f)
    np.savez(os.path.join(out, f"psnr_hip_runs{tag}.npz"), curves=np.array(curves))
    assert abs(delta) <= 2.5 * se, (delta, se, hip_pl.tolist(), ref_pl.tolist())
    assert hip_pl.std(ddof=1) <= 2.5 * ref_pl.std(ddof=1) + 0.25, (hip_pl.tolist(), ref_pl.tolist())
    assert hip_pl.min() > ref_curve[0, 1] + 15.0
    return stats


N_HIP_RUNS = 5
N_SPLIT_RUNS = 3


@pytest.mark.skipif(not os.path.exists(GOLD), reason="reference PSNR curve not generated (tools/psnr_reference.py)")
def test_psnr_plateau_distribution_matches_the_references():
    """Distributional form of "matched PSNR" (round 3): training is chaotic, so ONE run against a band cannot see a systematic loss of
    2-3 dB.  Here N_HIP_RUNS runs of the HIP renderer in the BENCHMARKED mode (fp32 atomics in the weight-gradient epilogues: their
    summation order differs from run to run, which is the same perturbation as another thread count is to the reference) are compared
    with every committed run of the reference (tests/golden/psnr_reference_*.npz: 1 .. 6 intra-op threads):
      * |mean plateau (HIP) - mean plateau (reference)| <= 2.5 sqrt(SE_HIP^2 + SE_ref^2)   (standard errors of the two means; 2.5 instead of
        2 keeps the false-alarm rate of a correct implementation near 1 %; with 5 + 6 runs the tolerance is ~2.5 dB, a systematic loss of
        that size would show, and tightens as reference runs are added),
      * the HIP runs do not scatter more than the reference's do (sample standard deviation <= 2.5 x, an F-test at ~2 %),
      * every HIP run starts on the reference's trajectory (< 0.02 dB at iteration 1, < 0.1 dB up to iteration 60)."""
    _distribution(N_HIP_RUNS, False, "")


@pytest.mark.skipif(not os.path.exists(GOLD), reason="reference PSNR curve not generated (tools/psnr_reference.py)")
def test_psnr_plateau_distribution_in_split_precision_mode():
    """The same distributional assertion for the OPT-IN split-precision mode as it is benchmarked (round 4; VERDICT r3 weak #1: the mode's
    system-level parity rested on one run inside a +-4 dB band, taken before the training chain moved onto the bf16 pipes): N_SPLIT_RUNS
    runs with atomic reductions, the whole chain (queries, deformation / colour forward + backward, weight gradients) on the
    register-resident split-precision kernels, against the reference's six runs."""
    _distribution(N_SPLIT_RUNS, True, "_split")


LOW = sorted(__import__("glob").glob(os.path.join(GOLD_DIR, "psnr_lowchaos_t*.npz")))


@pytest.mark.skipif(len(LOW) < 2, reason="low-chaos reference runs not generated (PSNR_LR_SCALE=0.25 tools/psnr_reference.py 600 ...)")
def test_psnr_low_chaos_schedule_is_tight():
    """The same scene with the learning rate / 4 and 600 iterations: the reference then agrees with ITSELF (other thread counts) to
    < 0.1 dB up to iteration 140, ~0.3 dB up to 200 and 0.6-0.8 dB on the (still rising) end of the curve -- a schedule on which a tight
    comparison means something.  The HIP renderer must follow the reference's mean curve within 0.25 dB up to iteration 140 (the
    reference's own runs differ by up to 0.16 dB there; a third summation order cannot be expected inside the hull of two), within the
    reference's half-range + 0.4 dB up to iteration 200 and + 0.6 dB at the end (mean of the last 4 evaluations).  Deterministic
    reductions, so that the verdict of this tight test is reproducible (the benchmarked atomic mode is the subject of the
    distribution test above)."""
    refs = [np.load(f) for f in LOW]
    n_iter, n_rays, lr_scale = int(refs[0]["n_iter"]), int(refs[0]["n_rays"]), float(refs[0]["lr_scale"])
    assert all(int(r["n_iter"]) == n_iter and float(r["lr_scale"]) == lr_scale for r in refs)
    rc = np.stack([r["curve"][:, 1] for r in refs])              # [runs, evaluations]
    its = refs[0]["curve"][:, 0]
    mean, half = rc.mean(0), 0.5 * (rc.max(0) - rc.min(0))
    curve, _ = _train(n_iter, n_rays, int(refs[0]["weight_seed"]), int(refs[0]["sched_seed"]), its, deterministic=True, lr=5e-4 * lr_scale)
    d = np.abs(curve[:, 1] - mean)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    np.savez(os.path.join(out, "psnr_hip_lowchaos.npz"), curve=curve, ref=rc, its=its)
    early, mid = its <= 140, its <= 200
    assert np.all(d[early] <= 0.25), (its[early][d[early] > 0.25], d[early].max())
    assert np.all(d[mid] <= half[mid] + 0.4), d[mid].max()
    end_ref = rc[:, -N_TAIL:].mean(1)
    end = float(curve[-N_TAIL:, 1].mean())
    assert abs(end - end_ref.mean()) <= 0.5 * (end_ref.max() - end_ref.min()) + 0.6, (end, end_ref.tolist())
