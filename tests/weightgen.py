"""Build-owned deterministic weight generator (numpy PCG64, stable across machines).

The same arrays are loaded into the reference (tools/make_golden.py, via
load_state_dict), into the oracle and into the HIP renderer, so golden fixtures only
need to store rays + outputs, not 6.6 MB of weights.

Key names / shapes follow the reference state_dict (endosurf.py:559-568).
"""
import numpy as np

DEFORM_DIMS = [(52, 256), (256, 256), (256, 256), (256, 204), (256, 256), (256, 256), (256, 256), (256, 256), (256, 3)]
SDF_DIMS = [(39, 256), (256, 256), (256, 256), (256, 256), (295, 256), (256, 256), (256, 256), (256, 256), (256, 257)]
COLOR_DIMS = [(349, 256), (256, 256), (256, 256), (256, 256), (605, 256), (256, 256), (256, 256), (256, 256), (256, 3)]
NET_DIMS = {"deform_network": DEFORM_DIMS, "sdf_network": SDF_DIMS, "color_network": COLOR_DIMS}


def _default_linear(rng, fin, fout):
    bound = 1.0 / np.sqrt(fin)
    W = rng.uniform(-bound, bound, size=(fout, fin))
    b = rng.uniform(-bound, bound, size=(fout,))
    return W, b


def _geometric_sdf(rng, l, fin, fout, bias=0.8, in_dim=39):
    # mirrors the distributions of build_mlp_nerf's geometric init (utils.py:36-56)
    if l == 8:
        W = rng.normal(np.sqrt(np.pi) / np.sqrt(fin), 1e-4, size=(fout, fin))
        b = np.full((fout,), -bias)
    elif l == 0:
        W = np.zeros((fout, fin))
        W[:, :3] = rng.normal(0.0, np.sqrt(2) / np.sqrt(fout), size=(fout, 3))
        b = np.zeros((fout,))
    elif l == 4:
        W = rng.normal(0.0, np.sqrt(2) / np.sqrt(fout), size=(fout, fin))
        W[:, -(in_dim - 3):] = 0.0
        b = np.zeros((fout,))
    else:
        W = rng.normal(0.0, np.sqrt(2) / np.sqrt(fout), size=(fout, fin))
        b = np.zeros((fout,))
    return W, b


def make_state(seed: int, mode: str = "init", use_deform: bool = True):
    """Return {full_key: float32 ndarray}; mode in {"init", "trained"}."""
    rng = np.random.default_rng(seed)
    out = {}
    for net, dims in NET_DIMS.items():
        for l, (fin, fout) in enumerate(dims):
            if net == "sdf_network":
                W, b = _geometric_sdf(rng, l, fin, fout)
            else:
                W, b = _default_linear(rng, fin, fout)
            g = np.linalg.norm(W, axis=1, keepdims=True)
            v = W.copy()
            if mode == "trained":
                # break the init structure (dense high-frequency columns, re-scaled rows, biases)
                # while keeping a zero level set inside the unit sphere so rays still hit a surface
                is_sdf = net == "sdf_network"
                g = g * np.exp((0.08 if is_sdf else 0.25) * rng.normal(size=g.shape))
                amp = 0.06 if is_sdf else 0.15
                v = v + (amp * np.abs(v).mean() + (0.004 if is_sdf else 0.02) / np.sqrt(fin)) * rng.normal(size=v.shape)
                b = b + (0.01 if is_sdf else 0.03) * rng.normal(size=b.shape)
                if is_sdf and l == 8:
                    g[0] = np.linalg.norm(W[0]) * 1.05
                    b[0] = -0.78
                if net == "deform_network" and l == 8:
                    g = g * 3.0
                if net == "color_network" and l == 8:
                    g = g * 25.0
            elif mode != "init":
                raise ValueError(mode)
            if net == "deform_network" and not use_deform:
                continue
            out[f"{net}.net.{l}.bias"] = b.astype(np.float32)
            out[f"{net}.net.{l}.weight_g"] = g.astype(np.float32)
            out[f"{net}.net.{l}.weight_v"] = v.astype(np.float32)
    var = 0.3 if mode == "init" else 0.36
    out["deviation_network.variance"] = np.array(var, dtype=np.float32)
    return out


def make_rays(seed: int, n: int, t: float = None, jitter: float = 0.01, spread: float = 1.0):
    """Synthetic camera of SURVEY 8d: o=(0,0,-1.5)+N(0,jitter^2), pinhole 640x512 f=800,
    d = normalize((u-cx)/f,(v-cy)/f,1); one time value per batch. Returns [n,9] float32."""
    rng = np.random.default_rng(seed)
    u = rng.uniform(0, 639, size=n) * spread + 319.5 * (1 - spread)
    v = rng.uniform(0, 511, size=n) * spread + 255.5 * (1 - spread)
    d = np.stack([(u - 319.5) / 800.0, (v - 255.5) / 800.0, np.ones(n)], -1)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.array([0.0, 0.0, -1.5])[None] + jitter * rng.normal(size=(n, 3))
    tt = rng.uniform() if t is None else t
    rays = np.concatenate([o, d, np.zeros((n, 2)), np.full((n, 1), tt)], -1)
    return rays.astype(np.float32)


def make_targets(seed: int, n: int):
    """colour U[0,1)^3, depth 1.2+0.2U, masks = 1."""
    rng = np.random.default_rng(seed + 7919)
    return dict(color=rng.uniform(size=(n, 3)).astype(np.float32),
                depth=(1.2 + 0.2 * rng.uniform(size=(n, 1))).astype(np.float32),
                mask=np.ones((n, 1), np.float32), color_mask=np.ones((n, 1), np.float32))
