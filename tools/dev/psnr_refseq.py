#!/usr/bin/env python3
"""PSNR plateau of complete 1500-iteration trainings through the reference trainer's OWN call sequence (three renderer calls,
torch.optim.Adam; the auxiliary calls' points in the render workspace's tail, errorondepth deferred) against the committed runs of the
reference (tests/golden/psnr_reference_*.npz): -> gpurun_out/psnr_refseq.json (copied to profiles/ by hand)."""
import json, os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_psnr as T
g = np.load(T.GOLD)
n_iter, n_rays, ref_curve = int(g["n_iter"]), int(g["n_rays"]), g["curve"]
runs = T._reference_runs()
ref_pl = np.array([float(np.mean(c[-T.N_TAIL:, 1])) for c in runs.values()])
hip = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    keep = []
    curve, _ = T._train(n_iter, n_rays, int(g["weight_seed"]), int(g["sched_seed"]), ref_curve[:, 0], deterministic=False, reference_sequence=True, keep=keep)
    hip.append(float(np.mean(curve[-T.N_TAIL:, 1])))
    print("run", k, "plateau", hip[-1], "tails", keep[0].tails_made, flush=True)
hip = np.array(hip)
se = float(np.sqrt(hip.var(ddof=1) / len(hip) + ref_pl.var(ddof=1) / len(ref_pl)))
out = dict(mode="reference call sequence (renderer(rays) -> errorondepth -> surface_neighbour_error, torch.optim.Adam), fp32, atomic reductions",
           reference_plateaus_db=ref_pl.tolist(), hip_plateaus_db=hip.tolist(), delta_mean_db=float(hip.mean() - ref_pl.mean()), standard_error_db=se,
           hip_std_db=float(hip.std(ddof=1)), reference_std_db=float(ref_pl.std(ddof=1)), n_iter=n_iter, n_rays=n_rays)
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(R, "gpurun_out", "psnr_refseq.json"), "w"), indent=1)
print(json.dumps(out))
