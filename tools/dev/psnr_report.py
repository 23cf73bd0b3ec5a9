#!/usr/bin/env python3
"""Print the PSNR-parity table from the committed reference curves (tests/golden/psnr_reference_long.npz, psnr_reference_t3.npz)
and the HIP curves a GPU run of tests/test_gpu_psnr.py left in gpurun_out/ (psnr_hip.npz, psnr_hip_split.npz)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = os.path.join(REPO, "tests", "golden")
O = os.path.join(REPO, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
ref = np.load(os.path.join(G, "psnr_reference_long.npz"))["curve"]
import glob
cols = {"reference fp32 (4 threads)": ref}
extra = [(f"reference fp32 ({os.path.basename(f)[16:-4]} threads)", f) for f in sorted(glob.glob(os.path.join(G, "psnr_reference_t*.npz")))]
for name, path in extra + [("HIP fp32 (deterministic)", os.path.join(O, "psnr_hip.npz")), ("HIP split precision", os.path.join(O, "psnr_hip_split.npz"))]:
    if os.path.exists(path):
        cols[name] = np.load(path)["curve"]
its = [1, 30, 60, 100, 150, 200, 250, 300, 400, 500, 600, 800, 1000, 1200, 1350, 1400, 1450, 1500]
print("| iteration | " + " | ".join(cols) + " |")
print("|---|" + "---|" * len(cols))
for it in its:
    row = []
    for c in cols.values():
        m = c[c[:, 0] == it]
        row.append(f"{m[0, 1]:.2f}" if len(m) else "-")
    print(f"| {it} | " + " | ".join(row) + " |")
print("| mean of the last 4 evaluations | " + " | ".join(f"{np.mean(c[-4:, 1]):.2f}" if c[-1, 0] == ref[-1, 0] else "-" for c in cols.values()) + " |")
for name, c in cols.items():
    n = min(len(c), len(ref))
    print(f"max |difference to the first column| over {n} evaluations: {name}: {np.max(np.abs(c[:n, 1] - ref[:n, 1])):.2f} dB")
