#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation (container only).

Usage (in the build container, where /root/reference exists):
    python tools/make_golden.py            # writes tests/golden/*.npz

The reference renderer (src/renderer/endosurf.py) is imported unmodified with two stub
modules for dependencies that are not installed (``mcubes`` and ``src.trainer.utils``; the
hot path uses neither).  It is driven with build-owned deterministic weights
(tests/weightgen.py) and rays; only inputs and the reference's outputs are stored.
Nothing from /root/reference is copied; this script and the fixtures are what travels.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np
import torch
import yaml

import weightgen  # noqa: E402

REF_ROOT = os.environ.get("ENDOSURF_REFERENCE", "/root/reference")


def import_reference():
    sys.modules["mcubes"] = types.ModuleType("mcubes")
    stub = types.ModuleType("src.trainer.utils")
    stub.tensor2array = lambda t: t.detach().cpu().numpy()
    tr = types.ModuleType("src.trainer")
    tr.__path__ = []
    sys.modules["src.trainer"] = tr
    sys.modules["src.trainer.utils"] = stub
    os.chdir(REF_ROOT)
    sys.path.insert(0, REF_ROOT)
    import src.renderer.endosurf as E
    return E


def load_cfg(use_deform):
    cfg = yaml.safe_load(open(os.path.join(REF_ROOT, "configs/endosurf/baseline/base_pull.yml")))
    cfg["net"]["use_deform"] = use_deform
    return cfg


def build_ref(E, cfg, state):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = E.EndoSurfRenderer(cfg["render"], cfg["net"], device="cpu")
    nets = ["sdf_network", "color_network", "deviation_network"] + (["deform_network"] if cfg["net"]["use_deform"] else [])
    ckpt = {}
    for net in nets:
        ckpt[net] = {k[len(net) + 1:]: torch.from_numpy(np.array(v)) for k, v in state.items() if k.startswith(net + ".")}
    r.load_checkpoint(ckpt)
    return r


def z_trace(r, rays, iter_step, u_perturb):
    """Drive the reference's own up_sample / cat_z_vals to expose intermediate z_vals."""
    n = rays.shape[0]
    o, d, time = rays[:, :3], rays[:, 3:6], rays[:, 8]
    import src.renderer.utils as U
    near, far, _ = U.get_sphere_intersection(o, d)
    t_vals = torch.linspace(0.0, 1.0, r.n_samples)
    z = near + (far - near) * t_vals[None, :]
    if u_perturb is not None:
        z = z + (u_perturb - 0.5) * (2.0 / r.n_samples)
    trace = [z.clone()]
    sdfs = []
    with torch.no_grad():
        dz = d / (d[..., 2:] + 1e-6)
        pts = (o[:, None, :] + dz[:, None, :] * z[..., :, None]).reshape(-1, 3)
        t = time[..., None, None].expand(n, r.n_samples, 1).reshape(-1, 1)
        sdf = r.model.get_sdf_from_observed_space(pts, t).reshape(n, r.n_samples)
        sdfs.append(sdf.clone())
        for i in range(r.up_sample_steps):
            new_z = r.up_sample(o, d, z, sdf, r.n_importance // r.up_sample_steps, 64 * 2 ** i)
            z, sdf = r.cat_z_vals(o, d, time, z, new_z, sdf, last=(i + 1 == r.up_sample_steps))
            trace.append(z.clone())
            sdfs.append(sdf.clone())
    return near, far, trace, sdfs


def grad_summary(named_params, rng_seed=1234, n_samp=32):
    out = {}
    rng = np.random.default_rng(rng_seed)
    for name, p in named_params:
        g = p.grad.detach().numpy().astype(np.float64).reshape(-1)
        idx = rng.integers(0, g.size, size=min(n_samp, g.size))
        out[f"grad/{name}/norm"] = np.array(np.linalg.norm(g))
        out[f"grad/{name}/sum"] = np.array(g.sum())
        out[f"grad/{name}/idx"] = idx.astype(np.int64)
        out[f"grad/{name}/val"] = g[idx].astype(np.float32)
    return out


def named_model_params(r):
    # names in the flat "net.key" form used by weightgen / the oracle
    return [(k.replace("model.", "", 1), p) for k, p in r.named_parameters()]


class patched_rng:
    """Feed explicit uniform numbers to the reference's torch.rand / torch.rand_like calls
    (endosurf.py:81 and :331) so fp32 and fp64 runs and the oracle all see the same draws."""

    def __init__(self, u_perturb, u_neigh_valid):
        self.u_perturb, self.u_neigh_valid = u_perturb, u_neigh_valid

    def __enter__(self):
        self._rand, self._rand_like = torch.rand, torch.rand_like
        up, un = self.u_perturb, self.u_neigh_valid

        def rand(*a, **k):
            assert up is not None and list(a[0]) == list(up.shape), (a, k)
            return up.clone()

        def rand_like(x, **k):
            assert un is not None and x.shape == un.shape, (x.shape, None if un is None else un.shape)
            return un.clone().to(x.dtype)

        torch.rand, torch.rand_like = rand, rand_like
        return self

    def __exit__(self, *exc):
        torch.rand, torch.rand_like = self._rand, self._rand_like


def full_grads(named_params):
    return {name: p.grad.detach().numpy().astype(np.float64).reshape(-1).copy() for name, p in named_params}


def run_reference(E, cfg, state, dtype, rays, tg, x, dd, tt, iter_step, u_perturb, u_neigh_full, scal_w, tag, out, full=None):
    """Everything captured from one reference instance at one dtype; keys get suffix ``tag``.  ``full`` (a dict) receives the complete
    parameter gradients of the two scalars ("grad", "scalgrad"), kept by the caller only to form whole-tensor error summaries."""
    torch.set_default_dtype(dtype)
    r = build_ref(E, cfg, state)
    if dtype == torch.float64:
        r = r.double()
        r.dtype = torch.float64
    cv = lambda a: a.to(dtype) if a is not None else None
    rays, x, dd, tt, u_perturb, u_neigh_full = map(cv, (rays, x, dd, tt, u_perturb, u_neigh_full))
    tg = {k: cv(v) for k, v in tg.items()}
    scal_w = [cv(w) for w in scal_w]
    use_deform = cfg["net"]["use_deform"]
    perturb = u_perturb is not None
    n_rays = rays.shape[0]
    npy = lambda t: t.detach().numpy()

    # ---- per-point network goldens ------------------------------------------------
    m = r.model
    with torch.no_grad():
        if use_deform:
            out[f"pt{tag}/deform"] = npy(m.deform_network(x, tt))
        out[f"pt{tag}/sdf_observed"] = npy(m.get_sdf_from_observed_space(x, tt))
    out[f"pt{tag}/J"] = npy(m.get_deform_grad_from_observed_space(x.clone(), tt))
    x_c = (x + torch.from_numpy(out[f"pt{tag}/deform"])) if use_deform else x
    with torch.no_grad():
        h = m.sdf_network(x_c)
    out[f"pt{tag}/sdf"] = npy(h[:, :1]); out[f"pt{tag}/feat"] = npy(h[:, 1:])
    out[f"pt{tag}/g_c"] = npy(m.get_sdf_grad_from_canonical_space(x_c.clone()))
    out[f"pt{tag}/g_o"] = npy(m.get_sdf_grad_from_observed_space(x.clone(), tt))
    out[f"pt{tag}/rgb"] = npy(m.forward(torch.cat([x, dd, tt], -1)))[:, 1:4]

    # ---- sampling trace + render_rays ----------------------------------------------
    near, far, trace, sdfs = z_trace(r, rays, iter_step, u_perturb)
    out[f"near{tag}"] = npy(near); out[f"far{tag}"] = npy(far)
    for i, z in enumerate(trace):
        out[f"z_trace{tag}/{i}"] = npy(z)
    for i, s_ in enumerate(sdfs[:-1]):
        out[f"sdf_trace{tag}/{i}"] = npy(s_)
    with patched_rng(u_perturb, None):
        ret = r(rays, iter_step=iter_step, perturb_overwrite=perturb)
    for k, v in ret.items():
        out[f"render{tag}/{k}"] = npy(v)

    # ---- errorondepth / ray_marching / surface_neighbour_error -----------------------
    se, ae, inside = r.errorondepth(rays, tg["depth"], tg["mask"])
    out[f"eod{tag}/sdf_error"] = npy(se); out[f"eod{tag}/angle_error"] = npy(ae); out[f"eod{tag}/inside"] = npy(inside)
    with torch.no_grad():
        d_i = r.ray_marching(rays, max_points=r.net_chunk)
    out[f"march{tag}/d_i"] = npy(d_i)
    valid = ((d_i.abs() != np.inf) & (d_i != 0) & (tg["mask"] == 1))[:, 0]
    n_valid = int(valid.sum())
    out[f"march{tag}/n_valid"] = np.array(n_valid)
    u_neigh_valid = u_neigh_full[valid]
    with patched_rng(None, u_neigh_valid):
        sn = r.surface_neighbour_error(rays=rays, mask=tg["mask"], neighbour_rad=0.1)
    out[f"sn{tag}/value"] = (npy(sn) if torch.is_tensor(sn) else np.array(sn, np.float32))

    # ---- full training loss + parameter gradients (trainer_endosurf.py:106-162 arithmetic) -----
    for p in r.parameters():
        p.grad = None
    with patched_rng(u_perturb, u_neigh_valid):
        ret = r(rays, iter_step=iter_step, perturb_overwrite=perturb)
        color_error = (ret["color_map"] - tg["color"]) * tg["color_mask"]
        color_loss = color_error.abs().sum() / (tg["color_mask"].sum() + 1e-10)
        sdf_loss, angle_loss, valid_depth = r.errorondepth(rays, d_gt=tg["depth"], mask=tg["mask"], iter_step=iter_step)
        depth_error = (ret["depth_map"] - tg["depth"]) * valid_depth * tg["mask"]
        depth_loss = depth_error.abs().sum() / ((valid_depth * tg["mask"]).sum() + 1e-10)
        eik = ret["gradient_o_error"]
        sn = r.surface_neighbour_error(rays=rays, mask=tg["mask"], iter_step=iter_step, neighbour_rad=0.1)
    loss = color_loss * 1.0 + depth_loss * 1.0 + sdf_loss * 1.0 + angle_loss * 0.1 + eik * 0.1 + 0.1 * sn
    loss.backward()
    for k, v in dict(total=loss, color=color_loss, depth=depth_loss, sdf=sdf_loss, angle=angle_loss, eikonal=eik,
                     surf_neig=sn).items():
        out[f"loss{tag}/{k}"] = npy(v) if torch.is_tensor(v) else np.array(v, np.float32)
    out.update({k.replace("grad/", f"grad{tag}/"): v for k, v in grad_summary(named_model_params(r)).items()})
    if full is not None:
        full["grad"] = full_grads(named_model_params(r))

    # ---- a render-only scalar with dense output weights (exercises every output) ----------------
    for p in r.parameters():
        p.grad = None
    with patched_rng(u_perturb, None):
        ret = r(rays, iter_step=iter_step, perturb_overwrite=perturb)
    cw, dw, gw, ww = scal_w
    scal = ((ret["color_map"] * cw).sum() + (ret["depth_map"] * dw).sum() + (ret["gradients_o"] * gw).sum()
            + (ret["weights"] * ww).sum() + 0.5 * ret["gradient_o_error"] + (ret["cdf"] * ww).sum() * 0.1
            + ret["s_val"].sum() * 0.01)
    scal.backward()
    out[f"scal{tag}/value"] = npy(scal)
    out.update({k.replace("grad/", f"scalgrad{tag}/"): v for k, v in grad_summary(named_model_params(r)).items()})
    if full is not None:
        full["scalgrad"] = full_grads(named_model_params(r))
    torch.set_default_dtype(torch.float32)
    return float(loss), n_valid


def make_case(E, name, seed, mode, use_deform, n_rays, iter_step, perturb):
    torch.set_default_dtype(torch.float32)
    cfg = load_cfg(use_deform)
    state = weightgen.make_state(seed, mode, use_deform)
    rays = torch.from_numpy(weightgen.make_rays(seed + 1, n_rays))
    tg = {k: torch.from_numpy(v) for k, v in weightgen.make_targets(seed + 2, n_rays).items()}
    rng = np.random.default_rng(seed + 3)
    tg["mask"] = torch.from_numpy((rng.uniform(size=(n_rays, 1)) > 0.2).astype(np.float32))   # non-trivial masks
    tg["color_mask"] = torch.from_numpy((rng.uniform(size=(n_rays, 1)) > 0.1).astype(np.float32))
    out = {"meta/seed": np.array(seed), "meta/iter_step": np.array(iter_step), "meta/use_deform": np.array(use_deform),
           "meta/mode": np.array(mode), "rays": rays.numpy()}
    for k, v in tg.items():
        out[f"target/{k}"] = v.numpy()
    M = 192
    x = torch.from_numpy((rng.uniform(-0.7, 0.7, size=(M, 3))).astype(np.float32))
    dd = rng.normal(size=(M, 3)); dd /= np.linalg.norm(dd, axis=-1, keepdims=True)
    dd = torch.from_numpy(dd.astype(np.float32))
    tt = torch.from_numpy(rng.uniform(size=(M, 1)).astype(np.float32))
    out.update({"pt/x": x.numpy(), "pt/d": dd.numpy(), "pt/t": tt.numpy()})
    u_perturb = torch.from_numpy(rng.uniform(size=(n_rays, 1)).astype(np.float32)) if perturb else None
    u_neigh = torch.from_numpy(rng.uniform(size=(n_rays, 3)).astype(np.float32))
    if perturb:
        out["u_perturb"] = u_perturb.numpy()
    out["u_neigh"] = u_neigh.numpy()
    scal_w = [torch.from_numpy(rng.normal(size=(n_rays, 3)).astype(np.float32)),
              torch.from_numpy(rng.normal(size=(n_rays, 1)).astype(np.float32)),
              torch.from_numpy((0.01 * rng.normal(size=(n_rays, 64, 3))).astype(np.float32)),
              torch.from_numpy((0.1 * rng.normal(size=(n_rays, 64))).astype(np.float32))]
    out.update({"scal/cw": scal_w[0].numpy(), "scal/dw": scal_w[1].numpy(), "scal/gw": scal_w[2].numpy(),
                "scal/ww": scal_w[3].numpy()})
    args = (rays, tg, x, dd, tt, iter_step, u_perturb, u_neigh, scal_w)
    f32, f64 = {}, {}
    loss32, nv32 = run_reference(E, cfg, state, torch.float32, *args, "", out, f32)       # the reference as shipped (fp32)
    loss64, nv64 = run_reference(E, cfg, state, torch.float64, *args, "64", out, f64)     # same code in fp64 = noise-free pin
    # the reference's OWN fp32-vs-fp64 relative L2 error of every parameter-gradient tensor, over the WHOLE tensor (round 4): the 32
    # sampled entries per tensor stored above can miss where that error sits (init_deform, colour layer 6: 0.015 % on the samples,
    # 0.69 % on the tensor), and the render-level gradient tests budget the HIP result against 3x this number
    for which in ("grad", "scalgrad"):
        for pname, g64 in f64[which].items():
            out[f"{which}err/{pname}/rel"] = np.array(np.linalg.norm(f32[which][pname] - g64) / (np.linalg.norm(g64) + 1e-300))
    # fp64 arrays are stored as float32 where that loses nothing relevant, to keep fixtures small
    for k in list(out):
        if out[k].dtype == np.float64 and out[k].size > 64:
            out[k] = out[k].astype(np.float32)
    path = os.path.join(REPO, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, loss32={loss32:.6f} loss64={loss64:.6f} n_valid={nv32}/{nv64}")


if __name__ == "__main__":
    E = import_reference()
    torch.set_num_threads(8)
    make_case(E, "init_deform", seed=101, mode="init", use_deform=True, n_rays=48, iter_step=1, perturb=False)
    make_case(E, "trained_deform", seed=202, mode="trained", use_deform=True, n_rays=48, iter_step=30000, perturb=True)
    make_case(E, "trained_nodeform", seed=303, mode="trained", use_deform=False, n_rays=48, iter_step=60000, perturb=False)
