#!/usr/bin/env python3
"""Golden vectors for ColorNetwork.forward(x, n, d, geo_feat) on EXPLICIT inputs (reference src/renderer/endosurf.py:828-842),
produced by running the REFERENCE implementation (build container only; import recipe in tools/make_golden.py).

    python tools/make_golden_color.py      # writes tests/golden/color_direct.npz

Inputs are build-owned (numpy PCG64): positions in the unit cube, un-normalised normals, view directions that are deliberately NOT
unit length (the colour network takes d as given), features at the scale the SDF network emits; weights from tests/weightgen.py.
Only inputs and the reference's fp32 / fp64 outputs are stored."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import make_golden as MG
import weightgen  # noqa: E402  (tests/ is put on sys.path by make_golden)

M = 200          # not a multiple of the 64-point tile


def inputs(seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1.0, 1.0, size=(M, 3))
    n = rng.normal(size=(M, 3)) * rng.uniform(0.2, 3.0, size=(M, 1))
    d = rng.normal(size=(M, 3))
    d = d / np.linalg.norm(d, axis=-1, keepdims=True) * rng.uniform(0.5, 1.5, size=(M, 1))
    feat = rng.normal(size=(M, 256)) * 0.5
    return dict(x=x.astype(np.float32), n=n.astype(np.float32), d=d.astype(np.float32), feat=feat.astype(np.float32))


if __name__ == "__main__":
    E = MG.import_reference()
    torch.set_num_threads(4)
    out = {}
    for name, seed, mode, use_deform in (("trained_deform", 202, "trained", True), ("init_deform", 101, "init", True)):
        state = weightgen.make_state(seed, mode, use_deform)
        inp = inputs(seed + 77)
        for k, v in inp.items():
            out[f"{name}/{k}"] = v
        out[f"{name}/meta"] = np.array([seed, int(mode == "trained"), int(use_deform)])
        for dtype, tag in ((torch.float32, "rgb"), (torch.float64, "rgb64")):
            torch.set_default_dtype(dtype)
            r = MG.build_ref(E, MG.load_cfg(use_deform), state)
            if dtype == torch.float64:
                r = r.double()
            with torch.no_grad():
                rgb = r.model.color_network(*(torch.from_numpy(inp[k]).to(dtype) for k in ("x", "n", "d", "feat")))
            out[f"{name}/{tag}"] = rgb.numpy()
            torch.set_default_dtype(torch.float32)
    path = os.path.join(MG.REPO, "tests", "golden", "color_direct.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")
