"""Thin training-step counterpart of the reference's EndoSurfTrainer (src/trainer/trainer_endosurf.py:94-203): the loss
arithmetic, Adam and the warm-up + cosine LR schedule around the drop-in renderer.  Host orchestration only (PyTorch-ROCm);
no logging / host synchronisation inside a step (the reference's 12 ``.item()`` calls per step live in its logger)."""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

LOSS_WEIGHTS = dict(color=1.0, depth=1.0, sdf=1.0, angle=0.1, eikonal=0.1, surf_neig=0.1)   # base_pull.yml:23-29


def compute_loss(renderer, batch: Dict[str, torch.Tensor], iter_step: int, weights=LOSS_WEIGHTS, surf_neig_rad: float = 0.1,
                 u_perturb=None, u_neigh=None):
    """compute_loss (trainer_endosurf.py:106-162) without the TensorBoard calls. Returns (total, terms, render dict)."""
    rays, color_gt, depth_gt = batch["rays"], batch["color"], batch["depth"]
    mask_gt, cmask = batch["mask"], batch["color_mask"]
    ret = renderer(rays, iter_step=iter_step, u_perturb=u_perturb)
    color_error = (ret["color_map"] - color_gt) * cmask
    color_loss = color_error.abs().sum() / (cmask.sum() + 1e-10)
    sdf_loss, angle_loss, valid = renderer.errorondepth(rays, d_gt=depth_gt, mask=mask_gt, iter_step=iter_step)
    depth_error = (ret["depth_map"] - depth_gt) * valid * mask_gt
    depth_loss = depth_error.abs().sum() / ((valid * mask_gt).sum() + 1e-10)
    eik = ret["gradient_o_error"]
    sn = renderer.surface_neighbour_error(rays=rays, mask=mask_gt, iter_step=iter_step, neighbour_rad=surf_neig_rad, u_neigh=u_neigh)
    total = (color_loss * weights["color"] + depth_loss * weights["depth"] + sdf_loss * weights["sdf"]
             + angle_loss * weights["angle"] + eik * weights["eikonal"] + weights["surf_neig"] * sn)
    terms = dict(color=color_loss, depth=depth_loss, sdf=sdf_loss, angle=angle_loss, eikonal=eik, surf_neig=sn)
    return total, terms, ret


def compute_loss_fused(renderer, batch: Dict[str, torch.Tensor], iter_step: int, weights=LOSS_WEIGHTS, surf_neig_rad: float = 0.1,
                       u_perturb=None, u_neigh=None, loss_kernel: bool = True, exact: bool = False, group=None):
    """Same loss as compute_loss, but the auxiliary points of errorondepth (N) and surface_neighbour_error (2N) are evaluated
    inside the render's kernel launches (endosurf_amd extension ``aux_points``) instead of two extra tiny point evaluations."""
    rays = renderer._rays32(batch["rays"])
    color_gt, depth_gt, mask_gt, cmask = batch["color"], batch["depth"], batch["mask"], batch["color_mask"]
    N = rays.shape[0]
    eng = renderer.engine
    need_p, need_n = renderer.perturb and u_perturb is None, u_neigh is None
    if need_p or need_n:
        # the step's random draws (stratified jitter [N], neighbour offsets [N,3]) in ONE library launch instead of two torch.rand's;
        # in a captured step the device-resident step counter selects the subsequence (Trainer.train_step_graph)
        u4 = eng.uniform(4 * N, getattr(renderer, "_rng_step_dev", None))
        if need_p:
            u_perturb = u4[:N]
        if need_n:
            u_neigh = u4[N:].view(N, 3)
    renderer._weights()          # weight-norm + packing once per step, with the autograd node (ray marching below is no_grad)
    # The 128-proposal ray-marching query fills the GPU; the 8 secant iterations after it and the hierarchical sampling
    # (coarse query + 3 dependent 8-sample queries) are independent chains of small, latency-bound launches: run the sampling
    # on a side stream WHILE the main stream iterates the secant (two throughput-bound kernels would only slow each other)
    main = torch.cuda.current_stream(rays.device)
    side = getattr(renderer, "_side_stream", None)
    if side is None:
        side = renderer._side_stream = torch.cuda.Stream(device=rays.device)
    # (starting the sampling chain together with the marching query, or on a high-priority stream, was measured and is slower: DESIGN 4)
    ms = renderer._march_begin(rays)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        z = renderer.sample_z(rays, iter_step, u_perturb=u_perturb, racing=True)      # (shorter tiles for its coarse query: it races the secant)
    d_i = renderer._march_refine(ms)
    aux_x, aux_t, valid_sn = renderer._train_aux_points(rays, depth_gt, mask_gt, d_i, surf_neig_rad, u_neigh)    # one launch
    eod_pts = aux_x[:N]
    main.wait_stream(side)
    if not torch.cuda.is_current_stream_capturing():      # (a captured step owns its memory pool: nothing is recycled between replays)
        z.record_stream(main)
    # (the split-precision training chain takes the auxiliary points in the same launches as well; evaluating them with the fp32
    # kernels on the side stream instead was measured and is no faster: a co-running launch breaks the whole-round fit of the main ones)
    ret = renderer(rays, iter_step=iter_step, aux_points=(aux_x, aux_t), z_vals=z)
    a_sdf, a_go = ret["aux_sdf"], ret["aux_gradients_o"]
    if loss_kernel:
        eik = ret["gradient_o_error"]
        exact_args = None
        if exact:
            # EXACT big-batch normalisers under data parallelism: the <= 6 denominators of the loss (colour mask, inside-sphere
            # mask for the sdf / angle terms, valid x mask for depth, valid surface hits, the eikonal term's relaxed-sphere count) are summed
            # over the ranks BEFORE the backward -- one 20-byte all-reduce; appending them to the gradient bucket cannot work, the gradient
            # depends on them through 1 / D_k per term -- and every term is scaled by the world size: the mean over the ranks of the
            # per-rank gradients is then exactly the gradient of the concatenated batch.
            import torch.distributed as dist
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            den = renderer.engine.empty(5)
            _LossFn.sums(renderer.engine, rays, eod_pts, mask_gt, cmask, valid_sn, den)
            den_local_eik = ret["eik_den"]            # sum relax + 1e-6 of THIS render (returned by it, not read from engine state)
            den[4:5].copy_(den_local_eik - 1e-6)
            if world > 1:
                dist.all_reduce(den, group=group)
            eik = eik * (float(world) * den_local_eik[0] / (den[4] + 1e-6))
            exact_args = (den, float(world))
        total, t = _LossFn.apply(ret["color_map"], ret["depth_map"], eik, a_sdf, a_go, renderer.engine, rays, eod_pts,
                                 color_gt, depth_gt, mask_gt, cmask, valid_sn, weights, exact_args)
        return total, dict(color=t[0], depth=t[1], sdf=t[2], angle=t[3], eikonal=t[4], surf_neig=t[5]), ret
    color_loss = ((ret["color_map"] - color_gt) * cmask).abs().sum() / (cmask.sum() + 1e-10)
    sdf_loss, angle_loss, valid = renderer._eod_loss(rays, eod_pts, mask_gt, a_sdf[:N], a_go[:N])
    depth_loss = ((ret["depth_map"] - depth_gt) * valid * mask_gt).abs().sum() / ((valid * mask_gt).sum() + 1e-10)
    eik = ret["gradient_o_error"]
    sn = renderer._sn_loss(a_go[N:], valid_sn)
    total = (color_loss * weights["color"] + depth_loss * weights["depth"] + sdf_loss * weights["sdf"]
             + angle_loss * weights["angle"] + eik * weights["eikonal"] + weights["surf_neig"] * sn)
    terms = dict(color=color_loss, depth=depth_loss, sdf=sdf_loss, angle=angle_loss, eikonal=eik, surf_neig=sn)
    return total, terms, ret


class _LossFn(torch.autograd.Function):
    """All six loss terms + total and their gradients in one HIP launch (es_train_loss)."""

    @staticmethod
    def sums(eng, rays, eod_pts, mask, cmask, valid_sn, out):
        """This rank's normalisers {sum cmask, sum inside, sum valid x mask, n_valid} -> out[0:4] (first launch of the exact mode)."""
        import ctypes as C
        from . import _lib
        N = rays.shape[0]
        f = lambda t: t.detach().to(torch.float32).contiguous()
        keep = [f(rays), f(eod_pts), f(mask), f(cmask), (valid_sn.view(torch.uint8) if valid_sn.dtype == torch.bool else valid_sn.to(torch.uint8)).contiguous()]
        z3, z1 = eng.zeros(3 * N, 3), eng.zeros(3 * N, 1)
        a = _lib.es_loss_args()
        for name, t in zip(("rays", "eod_pts", "mask", "cmask", "valid_sn"), keep):
            setattr(a, name, _lib.ptr(t))
        # the first pass reads every input of the kernel: give it defined (zero) renderer outputs and targets
        for name, t in (("color_map", z3), ("color_gt", z3), ("depth_map", z1), ("depth_gt", z1), ("aux_sdf", z1), ("aux_go", z3), ("eik", z1)):
            setattr(a, name, _lib.ptr(t))
        a.N = N
        a.den_out = _lib.ptr(out)
        _lib.check(eng.lib.es_train_loss(C.byref(a), eng.st()), "es_train_loss")

    @staticmethod
    def forward(ctx, color_map, depth_map, eik, aux_sdf, aux_go, eng, rays, eod_pts, color_gt, depth_gt, mask, cmask, valid_sn, w, exact=None):
        import ctypes as C
        from . import _lib
        N = rays.shape[0]
        if mask.numel() != N or cmask.numel() != N or valid_sn.numel() != N or aux_sdf.numel() != 3 * N:
            raise ValueError("es_train_loss expects per-ray masks [N,1] and 3N auxiliary points")
        f = lambda t: t.detach().to(torch.float32).contiguous()
        ins = [f(color_map), f(depth_map), f(eik).reshape(1), f(aux_sdf), f(aux_go), f(rays), f(eod_pts), f(color_gt), f(depth_gt), f(mask),
               f(cmask), (valid_sn.view(torch.uint8) if valid_sn.dtype == torch.bool else valid_sn.to(torch.uint8)).contiguous()]
        terms, total = eng.empty(8), eng.empty(1)
        # the five adjoints in ONE buffer (a non-unit seed of the backward pass scales them with one launch)
        gbuf = eng.empty(16 * N + 1)
        grads = [gbuf[0:3 * N].view(N, 3), gbuf[3 * N:4 * N].view(N, 1), gbuf[4 * N:4 * N + 1], gbuf[4 * N + 1:7 * N + 1].view(3 * N, 1),
                 gbuf[7 * N + 1:16 * N + 1].view(3 * N, 3)]
        a = _lib.es_loss_args()
        for name, t in zip(("color_map", "depth_map", "eik", "aux_sdf", "aux_go", "rays", "eod_pts", "color_gt", "depth_gt", "mask", "cmask",
                            "valid_sn"), ins):
            setattr(a, name, _lib.ptr(t))
        a.N = N
        a.w_color, a.w_depth, a.w_sdf, a.w_angle, a.w_eik, a.w_sn = (float(w[k]) for k in ("color", "depth", "sdf", "angle", "eikonal", "surf_neig"))
        for name, t in zip(("terms", "g_color", "g_depth", "g_eik", "g_aux_sdf", "g_aux_go"), [terms] + grads):
            setattr(a, name, _lib.ptr(t))
        if exact is not None:            # (global normalisers [4+], world): see compute_loss_fused
            a.den_global, a.world = _lib.ptr(exact[0]), float(exact[1])
        a.total_out = _lib.ptr(total)
        _lib.check(eng.lib.es_train_loss(C.byref(a), eng.st()), "es_train_loss")
        ctx.set_materialize_grads(False)
        ctx.grads, ctx.gbuf, ctx.eng, ctx.n = grads, gbuf, eng, N
        ctx.eik_shape = eik.shape
        ctx.mark_non_differentiable(terms)
        return total.reshape(()), terms

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        if g_total is None:
            return (None,) * 15
        eng, N = ctx.eng, ctx.n
        if g_total.data_ptr() == eng.ones1.data_ptr():       # the step's own seed (Trainer: loss.backward(gradient=engine.ones1)): d total = 1
            gc, gd, ge, gs, gg = ctx.grads
        else:
            from . import _lib
            out = eng.empty(16 * N + 1)
            gt = g_total.detach().to(torch.float32).reshape(1)
            _lib.check(eng.lib.es_scale(_lib.ptr(out), _lib.ptr(ctx.gbuf), 16 * N + 1, _lib.ptr(gt), eng.st()), "es_scale")
            gc, gd, ge, gs, gg = (out[0:3 * N].view(N, 3), out[3 * N:4 * N].view(N, 1), out[4 * N:4 * N + 1], out[4 * N + 1:7 * N + 1].view(3 * N, 1),
                                  out[7 * N + 1:16 * N + 1].view(3 * N, 3))
        return (gc, gd, ge.reshape(ctx.eik_shape), gs, gg, None, None, None, None, None, None, None, None, None, None)


def _lib_flags_save(renderer) -> int:
    """The point-evaluation flags of a grad-enabled render of this renderer (what its workspace budget is computed for)."""
    from . import _lib
    return (_lib.PF_DEFORM if renderer.use_deform else 0) | _lib.PF_SAVE


def lr_factor(it: int, n_iter: int = 100000, warm_up_end: int = 5000, alpha: float = 0.05) -> float:
    """update_learning_rate (trainer_endosurf.py:183-203)."""
    if it < warm_up_end:
        return it / warm_up_end
    prog = (it - warm_up_end) / (n_iter - warm_up_end)
    return (math.cos(math.pi * prog) + 1.0) * 0.5 * (1 - alpha) + alpha


def cal_psnr(a: torch.Tensor, b: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """cal_psnr (src/trainer/utils.py:340-353), kept on device (no host sync)."""
    if mask.dim() == a.dim() - 1:
        mask = mask[..., None]
    mask_sum = mask.sum() + 1e-10
    return 20.0 * torch.log10(1.0 / torch.sqrt(((a - b) ** 2 * mask).sum() / (mask_sum * 3.0)))


class SyntheticScene:
    """Build-owned synthetic stand-in for Dataset.get_train_batch_data_by_index (dataset.py:117-161): pinhole camera
    640x512, f = 800 px at o = (0,0,-1.5) looking down +z, one time value per batch; targets are uniform colours,
    depth 1.2 + 0.2 U, masks = 1 (SURVEY 8d / BASELINE.md 3). Everything is generated on the device."""

    def __init__(self, device, seed: int = 0, jitter: float = 0.01):
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.jitter = jitter

    def batch(self, n_rays: int) -> Dict[str, torch.Tensor]:
        g, dev = self.gen, self.device
        u = torch.rand(n_rays, generator=g, device=dev) * 639.0
        v = torch.rand(n_rays, generator=g, device=dev) * 511.0
        d = torch.stack([(u - 319.5) / 800.0, (v - 255.5) / 800.0, torch.ones_like(u)], -1)
        d = d / d.norm(dim=-1, keepdim=True)
        o = torch.tensor([0.0, 0.0, -1.5], device=dev)[None] + self.jitter * torch.randn(n_rays, 3, generator=g, device=dev)
        t = torch.rand(1, generator=g, device=dev).expand(n_rays, 1)
        rays = torch.cat([o, d, torch.zeros(n_rays, 2, device=dev), t], -1).contiguous()
        return dict(rays=rays, color=torch.rand(n_rays, 3, generator=g, device=dev),
                    depth=1.2 + 0.2 * torch.rand(n_rays, 1, generator=g, device=dev),
                    mask=torch.ones(n_rays, 1, device=dev), color_mask=torch.ones(n_rays, 1, device=dev))


    def frame(self, H: int = 512, W: int = 640, t: float = 0.5, row0: int = 0, rows: int = None) -> torch.Tensor:
        """Rays [rows, W, 9] of one full frame of the same pinhole camera (no jitter), rows row0..row0+rows."""
        dev = self.device
        rows = H - row0 if rows is None else rows
        v, u = torch.meshgrid(torch.arange(row0, row0 + rows, device=dev, dtype=torch.float32),
                              torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
        d = torch.stack([(u - 319.5) / 800.0, (v - 255.5) / 800.0, torch.ones_like(u)], -1)
        d = d / d.norm(dim=-1, keepdim=True)
        o = torch.tensor([0.0, 0.0, -1.5], device=dev).expand(rows, W, 3)
        return torch.cat([o, d, torch.zeros(rows, W, 2, device=dev), torch.full((rows, W, 1), float(t), device=dev)], -1).contiguous()


class FlatAdam:
    """torch.optim.Adam with the reference's settings (trainer_endosurf.py:65-71: one group, defaults) as ONE kernel launch over
    the renderer's flat parameter buffer (es_adam_step).  The gradient is the flat buffer produced by es_weightnorm_backward
    (the parameters' ``.grad`` are views of it) plus the scalar variance gradient; if gradients were accumulated or produced
    some other way they are gathered from ``.grad`` first.  ``param_groups[0]['lr']`` is read at every step like torch's."""

    def __init__(self, renderer, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8):
        self.model, self.eng = renderer.model, renderer.engine
        self.flat = self.model._flat
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps)]
        self.step_count = 0
        self.scalars_dev = None          # device [step_size, bc2_sqrt, grad_scale] of a captured step (set by Trainer.capture_graph)
        self._named = [(self.model._layout[k][0], p) for k, p in self.model.ordered_params()]
        self._var = self.model.deviation_network.variance
        self._var_off = int(self.model._layout["deviation_network.variance"][0])
        self._all = [p for _, p in self._named] + [self._var]

    def zero_grad(self, set_to_none: bool = True):
        for p in self._all:
            p.grad = None
        self.model._flat_grad = None

    def flat_grad(self, include_variance: bool = False) -> torch.Tensor:
        """The gradient in flat-buffer layout (variance slot filled only if ``include_variance``)."""
        g = self.model._flat_grad
        off0, p0 = self._named[0]
        offl, pl = self._named[-1]
        ok = (g is not None and p0.grad is not None and pl.grad is not None and p0.grad.data_ptr() == g.data_ptr() + 4 * off0
              and pl.grad.data_ptr() == g.data_ptr() + 4 * offl)
        if not ok:                      # accumulated / foreign gradients: gather .grad into a fresh flat buffer
            g = torch.zeros_like(self.flat)
            with torch.no_grad():
                pairs = [(g[off:off + p.numel()].view(p.shape), p.grad) for off, p in self._named if p.grad is not None]
                if pairs:
                    torch._foreach_copy_([a for a, _ in pairs], [b for _, b in pairs])
        if include_variance and self._var.grad is not None:
            with torch.no_grad():
                g[self._var_off:self._var_off + 1].copy_(self._var.grad.reshape(1))
        return g

    def step(self, grad: torch.Tensor = None, grad_scale: float = 1.0, variance_in_grad: bool = False):
        from . import _lib
        g = self.flat_grad() if grad is None else grad
        gv = None if (variance_in_grad or self._var.grad is None) else self._var.grad
        frozen = [(p, p.detach().clone()) for p in self._all if not p.requires_grad]      # the launch updates the whole buffer
        self.step_count += 1
        pg = self.param_groups[0]
        b1, b2 = pg["betas"]
        if self.scalars_dev is not None:      # captured step: (step_size, bc2_sqrt, grad_scale) live in device memory (Trainer.train_step_graph)
            _lib.check(self.eng.lib.es_adam_step_dev(_lib.ptr(self.flat), _lib.ptr(g), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                                     self.flat.numel(), b1, b2, pg["eps"], _lib.ptr(self.scalars_dev),
                                                     _lib.ptr(gv) if gv is not None else None, self._var_off, self.eng.st()), "es_adam_step_dev")
            if frozen:
                with torch.no_grad():
                    torch._foreach_copy_([p for p, _ in frozen], [v for _, v in frozen])
            self.model._epoch += 1
            return
        step_size = pg["lr"] / (1.0 - b1 ** self.step_count)
        bc2_sqrt = math.sqrt(1.0 - b2 ** self.step_count)
        _lib.check(self.eng.lib.es_adam_step(_lib.ptr(self.flat), _lib.ptr(g), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                             self.flat.numel(), b1, b2, pg["eps"], step_size, bc2_sqrt, float(grad_scale),
                                             _lib.ptr(gv) if gv is not None else None, self._var_off, self.eng.st()), "es_adam_step")
        if frozen:
            with torch.no_grad():
                torch._foreach_copy_([p for p, _ in frozen], [v for _, v in frozen])
        self.model._epoch += 1            # parameters changed behind torch's version counters: invalidate the packed weights

    # ---- checkpoint format: torch.optim.Adam's own (what the reference stores as ckpt["optimizer"], trainer_endosurf.py:76-92) ----
    def _train_param_slots(self):
        """[(flat offset, numel, shape)] in the reference optimiser's parameter order = get_train_params() order
        (trainer_endosurf.py:60-71: deform, sdf, colour networks layer by layer (bias, weight_g, weight_v), then the variance)."""
        groups = self.model.get_train_params()
        byid = {id(p): (off, p) for off, p in self._named}
        byid[id(self._var)] = (self._var_off, self._var)
        out = []
        for key in groups:
            for p in groups[key]:
                off, q = byid[id(p)]
                out.append((off, q.numel(), tuple(q.shape)))
        return out

    def state_dict(self):
        """A torch.optim.Adam state_dict ({"state": {i: {step, exp_avg, exp_avg_sq}}, "param_groups": [...]}) that the reference
        trainer's ``optimizer.load_state_dict`` accepts: the flat moment buffers are scattered to per-parameter tensors."""
        import copy
        slots = self._train_param_slots()
        state = {}
        if self.step_count > 0:
            for i, (off, n, shape) in enumerate(slots):
                state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=self.exp_avg[off:off + n].view(shape).clone(),
                                exp_avg_sq=self.exp_avg_sq[off:off + n].view(shape).clone())
        pg = copy.deepcopy(self.param_groups[0])
        # the param_group keys of the installed torch's own Adam (they differ between torch versions), with this optimiser's values
        tmpl = torch.optim.Adam([torch.zeros(1)], lr=pg["lr"], betas=tuple(pg["betas"]), eps=pg["eps"]).state_dict()["param_groups"][0]
        group = dict(tmpl, lr=pg["lr"], betas=tuple(pg["betas"]), eps=pg["eps"], params=list(range(len(slots))))
        group.update({k: v for k, v in pg.items() if k not in group})
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        """Accepts a torch.optim.Adam state_dict (reference checkpoints: ckpt["optimizer"]) or the flat round-1 format
        {"step", "exp_avg", "exp_avg_sq", "param_groups"}."""
        import copy
        if "state" not in sd:           # flat format
            self.step_count = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.param_groups = [dict(copy.deepcopy(g)) for g in sd["param_groups"]]
            return
        slots = self._train_param_slots()
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(slots):
            raise ValueError(f"optimizer state_dict has {[len(g['params']) for g in groups]} parameters per group; this model has one group "
                             f"of {len(slots)} (use_deform = {self.model.use_deform})")
        g = groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("FlatAdam implements the reference's plain Adam (no weight decay / amsgrad / maximize)")
        state = {int(k): v for k, v in sd["state"].items()}
        steps = set()
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        for pos, pid in enumerate(g["params"]):
            st = state.get(int(pid))
            if st is None:
                continue
            off, n, shape = slots[pos]
            if tuple(st["exp_avg"].shape) != shape:
                raise ValueError(f"optimizer state of parameter {pos} has shape {tuple(st['exp_avg'].shape)}, expected {shape}")
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}); FlatAdam keeps one step count")
        self.step_count = steps.pop() if steps else 0
        self.param_groups = [dict(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"])]


class Trainer:
    """zero_grad -> compute_loss -> backward -> (data-parallel gradient all-reduce) -> Adam  (train_step, trainer_endosurf.py:94-104)."""

    def __init__(self, renderer, lr: float = 5e-4, n_iter: int = 100000, warm_up_end: int = 5000, lr_alpha: float = 0.05,
                 loss_weights=LOSS_WEIGHTS, surf_neig_rad: float = 0.1, data_parallel: bool = False, fused: bool = True,
                 schedule: str = None, flat_adam: bool = True, force_collective: bool = False, exact_denominators: bool = False, group=None,
                 overlap_allreduce: bool = False):
        self.renderer = renderer
        self.group = group                                  # torch.distributed process group of the data-parallel ranks (None: the default group)
        self.force_collective = bool(force_collective)      # issue the gradient all-reduce even at world size 1 (RCCL smoke test)
        groups = renderer.get_train_params()
        self.params = [p for k in groups for p in groups[k]]
        # Adam with the reference's defaults (trainer_endosurf.py:70); on the GPU the single-kernel "fused" implementation
        # of the same update is used (81 parameter tensors -> one launch)
        if flat_adam:
            self.optimizer = FlatAdam(renderer, lr=lr)          # the same update as one launch over the flat parameter buffer
        else:
            try:
                self.optimizer = torch.optim.Adam(params=self.params, lr=lr, fused=self.params[0].is_cuda)
            except (TypeError, RuntimeError):
                self.optimizer = torch.optim.Adam(params=self.params, lr=lr)
        self.lr_init, self.n_iter, self.warm_up_end, self.lr_alpha = lr, n_iter, warm_up_end, lr_alpha
        self.loss_weights, self.surf_neig_rad = loss_weights, surf_neig_rad
        self.data_parallel = data_parallel
        # "fused": auxiliary points inside the render launches (default); "plain": the reference's call sequence
        schedule = schedule or ("fused" if fused else "plain")
        self.loss_fn = {"fused": compute_loss_fused, "plain": compute_loss}[schedule]
        # data parallel: all-reduce the loss normalisers before the backward, so that N ranks x B rays train exactly like one rank with
        # N x B rays (default: standard DDP semantics, per-rank ratios averaged)
        self.exact_denominators = bool(exact_denominators)
        self.allreduce_events = None        # a list: train_step appends a HIP-event pair around its gradient all-reduce
        # OPT-IN (data parallel, eager steps): the gradient all-reduce as a pipeline of buckets -- a bucket's weight-norm backward and its
        # all-reduce are issued on a side stream as soon as the weight-gradient launch that completes it is behind the main stream
        # (_pipeline_hook).  Same sums, same update; no multi-GPU box has measured it yet, so the default stays one bucket.
        self.overlap_allreduce = bool(overlap_allreduce)
        self._ar_stream = None
        self.pipelined_steps = 0            # steps whose gradient really went through the bucket pipeline (not its one-bucket fallback)
        if self.exact_denominators:
            if schedule != "fused":
                raise ValueError("exact_denominators needs the fused schedule")
            import functools
            self.loss_fn = functools.partial(compute_loss_fused, exact=True, group=group)

    def update_learning_rate(self, global_step: int):
        lr = self.lr_init * lr_factor(global_step, self.n_iter, self.warm_up_end, self.lr_alpha)
        for g in self.optimizer.param_groups:
            g["lr"] = lr
        return lr

    def train_step(self, batch, global_step: int, u_perturb=None, u_neigh=None):
        with torch.cuda.device(self.renderer.device):
            return self._train_step(batch, global_step, u_perturb, u_neigh)

    def save_checkpoint(self, global_step: int):
        """The reference's ckpt.tar dictionary (trainer_endosurf.py:85-92): the four network state_dicts, "n_iter", "optimizer"."""
        ckpt = self.renderer.save_checkpoint()
        ckpt["n_iter"] = int(global_step)
        ckpt["optimizer"] = self.optimizer.state_dict()
        return ckpt

    def load_checkpoint(self, ckpt) -> int:
        """Resume from a (reference or own) ckpt dictionary (trainer_endosurf.py:76-83); returns the next iteration."""
        self.renderer.load_checkpoint(ckpt)
        if "optimizer" in ckpt:
            self.optimizer.load_state_dict(ckpt["optimizer"])
        return int(ckpt.get("n_iter", 0)) + 1

    # ---- whole-step hipGraph ("so the whole training step is graph-capturable") -----------------------------------
    def train_step_graph(self, batch, global_step: int):
        """``train_step`` with the whole step -- weight-norm packing, ray marching, sampling, render, loss, backward, Adam: ~150 launches
        on two streams -- captured ONCE per batch shape in a hipGraph (torch.cuda.CUDAGraph) and replayed.  Everything that changes from
        step to step lives in device memory: the batch (copied into static buffers), the step counters from which the first launch of the
        step computes the learning rate, Adam's bias corrections and the cos-anneal ratio (es_train_schedule: the reference's
        update_learning_rate / get_cos_anneal_ratio on the device), the random draws (es_uniform: Philox keyed by torch.initial_seed(), the
        subsequence taken from the device-resident step counter).  The first two calls
        run eagerly (lazy initialisation; they are ordinary training steps), the third captures.  Under data parallelism the graph ends
        with the flat gradient; the all-reduce and the Adam launch follow it eagerly.  Returns the loss (a static device tensor)."""
        with torch.cuda.device(self.renderer.device):
            return self._train_step_graph(batch, global_step)

    def _graph_key(self, batch, global_step: int):
        """Everything the captured launch sequence depends on besides device memory: the batch shape and the HOST-evaluated branches of a
        step (importance sampling on / off at this iteration, ray-marching block mode, kernel family switches, deterministic reductions).
        A change re-captures instead of silently replaying the old branch."""
        r, e = self.renderer, self.renderer.engine
        upsample = global_step >= r.important_begin_iter and r.n_importance > 0
        return (tuple(batch["rays"].shape), tuple(sorted(batch)), bool(upsample), int(e.march_block), bool(e.split_precision), bool(e.x3_train_chain),
                int(e.x3_query_min), bool(e.deterministic), bool(self.data_parallel))

    def _train_step_graph(self, batch, global_step: int):
        r, opt = self.renderer, self.optimizer
        if not isinstance(opt, FlatAdam) or self.loss_fn is not compute_loss_fused:
            raise ValueError("train_step_graph needs the fused schedule and FlatAdam (the defaults; exact_denominators puts a collective "
                             "into the forward pass and is not captured)")
        g = getattr(self, "_graph", None)
        key = self._graph_key(batch, global_step)
        if g is None or g["key"] != key:
            scal = r.engine.zeros(4)          # step_size, bc2_sqrt, grad_scale | cos_anneal: written by es_train_schedule inside the step
            g = self._graph = dict(key=key, batch={k: torch.empty_like(v) for k, v in batch.items()}, scal=scal,
                                   state=torch.zeros(2, device=r.device, dtype=torch.float64), next=None, eager=0, graph=None, loss=None)
        world = 1
        if self.data_parallel:
            import torch.distributed as dist
            world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        for k, v in batch.items():
            g["batch"][k].copy_(v, non_blocking=True)
        t = opt.step_count + 1
        if g["next"] != (global_step, t):      # the device-resident counters follow the host's (first call, a jump in the step number)
            g["state"].copy_(torch.tensor([global_step - 1, t - 1], dtype=torch.float64))
        g["next"] = (global_step + 1, t + 1)
        pg = opt.param_groups[0]
        # the host's copy of the schedule follows the device's (same formula, both in double precision): checkpoints written from a
        # graph-mode run carry the learning rate of the step they were written at, like the reference's (trainer_endosurf.py:85-92)
        self.update_learning_rate(global_step)

        def schedule():
            from . import _lib
            _lib.check(r.engine.lib.es_train_schedule(_lib.ptr(g["state"]), float(self.lr_init), float(self.n_iter), float(self.warm_up_end),
                                                      float(self.lr_alpha), float(pg["betas"][0]), float(pg["betas"][1]), 1.0 / world,
                                                      float(r.anneal_end), _lib.ptr(g["scal"]), r.engine.st()), "es_train_schedule")

        def body():
            schedule()
            # (optional "u_perturb" / "u_neigh" entries of the batch replace the random draws: reproducible tests)
            loss, _, _ = self._step_body(g["batch"], global_step, g["batch"].get("u_perturb"), g["batch"].get("u_neigh"))
            if self.data_parallel:
                return loss.detach(), opt.flat_grad(include_variance=True)
            opt.step()
            return loss.detach(), None

        def finish(flat):
            if self.data_parallel:
                from .parallel import allreduce_flat
                allreduce_flat(flat, group=self.group, force=self.force_collective)
                opt.step(grad=flat, variance_in_grad=True)

        # The device-resident scalars are visible to the launches of THIS call only: an eager train_step / render / checkpoint afterwards
        # must see the host schedule again (pg["lr"], grad_scale, get_cos_anneal_ratio(iter_step)), not the last graph step's values.
        opt.scalars_dev, r._cos_anneal_dev = g["scal"], g["scal"][3:]
        r._rng_step_dev = g["state"]          # (the step counter selects the random subsequence of a replay: engine.uniform)
        try:
            if g["graph"] is None and g["eager"] < 2:        # lazy initialisation outside a capture: two ordinary steps
                g["eager"] += 1
                loss, flat = body()
                finish(flat)
                return loss
            if g["graph"] is None:
                count = opt.step_count
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    g["loss"], g["flat"] = body()
                opt.step_count = count                         # capturing launches nothing: the step count advances with the replays
                g["graph"] = graph
                # The step arena is the one buffer of a captured step that does NOT live in the graph's private pool: its address is baked
                # into the graph's memset and into every launch that uses a slice of it.  The graph keeps it alive, so an eager step with a
                # larger batch in between (which makes the engine allocate a new, larger arena) cannot hand the old block back to the
                # caching allocator while replays still write into it.
                g["arena"] = r.engine._arena
            g["graph"].replay()
            if self.data_parallel:
                finish(g["flat"])                              # (opt.step advances the count itself)
            else:
                opt.step_count = t
        finally:
            opt.scalars_dev, r._cos_anneal_dev = None, None
            r._rng_step_dev = None
        r.model._epoch += 1                                # parameters changed behind Python's back: packed weights are stale
        r.model._pack_cache = None
        return g["loss"]

    def _step_body(self, batch, global_step: int, u_perturb=None, u_neigh=None):
        """zero_grad -> loss -> backward inside the engine's step arena (one memset for all of the step's zero-initialised buffers) and
        with the engine's persistent ones as the backward seed (no fill launch; the loss node hands its adjoints on unscaled)."""
        eng = self.renderer.engine
        self.optimizer.zero_grad(set_to_none=True)
        eng.arena_begin(extra_floats=batch["rays"].shape[0] * 128)
        try:
            loss, terms, ret = self.loss_fn(self.renderer, batch, global_step, self.loss_weights, self.surf_neig_rad, u_perturb, u_neigh)
            loss.backward(gradient=eng.ones1.reshape(loss.shape) if loss.numel() == 1 and loss.dtype == torch.float32 else None)
        finally:
            eng.arena_end()
        return loss, terms, ret

    # ---- pipelined gradient all-reduce (opt-in) -----------------------------------------------------------------------------------
    def _bucket_plan(self):
        """Flat-buffer ranges and weight-norm layer ranges (index = 9 * network + layer) of the buckets.  The weight-gradient launches
        run deform -> sdf -> colour, and the deformation network's LAST layer is finished by slices riding in the other two launches:
          A  deform layers 0..7   complete behind the deform launch   (side stream, under the sdf + colour launches)
          B  sdf                  complete behind the sdf launch      (side stream, under the colour launch)
          C  deform layer 8, colour, variance                         (main stream, after the backward: exposed)"""
        plan = getattr(self, "_plan", None)
        if plan is None:
            from . import params as P
            m = self.renderer.model
            off = lambda ni, l: int(m._layout[f"{P.NET_NAMES[ni]}.net.{l}.bias"][0])
            n_param = int(self.renderer.engine.n_param)
            deform = bool(m.use_deform)
            plan = self._plan = dict(
                A=((0, 8), (off(0, 0), off(0, 8))) if deform else None,
                B=((9, 9), (off(1, 0), off(2, 0))),
                remaining=([(8, 1)] if deform else []) + [(18, 9)],
                C=([(off(0, 8), off(1, 0))] if deform else []) + [(off(2, 0), n_param)])
        return plan

    def _pipeline_hook(self, stage, dweff):
        """Called by Engine.point_backward behind every weight-gradient launch of the step (main stream)."""
        from . import _lib
        from .parallel import allreduce_flat
        eng = self.renderer.engine
        pipe = eng._grad_pipeline
        plan = self._bucket_plan()
        if pipe["dflat"] is None:
            # _PackFn.backward recognises the hooks' buffer by its address AND by the step having produced exactly ONE gradient buffer
            # of the effective weights (autograd sums the gradients of several consumers, possibly in place into the first one: same
            # address, other contents; Engine.point_backward counts the buffers in pipe["buffers"])
            pipe["dflat"], pipe["dweff_ptr"], pipe["remaining"] = eng.zeros(eng.n_param), dweff.data_ptr(), plan["remaining"]
        job = plan["A"] if stage == _lib.BWD_WGRAD_DEFORM else (plan["B"] if stage == _lib.BWD_WGRAD_SDF else None)
        if job is None:
            return
        (first, n), (a, b) = job
        main = torch.cuda.current_stream(eng.device)
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(self._ar_stream):
            self._ar_stream.wait_event(ev)
            eng.weightnorm_backward_layers(self.renderer.model._flat, dweff, pipe["dflat"], first, n)
            allreduce_flat(pipe["dflat"][a:b], group=self.group, force=self.force_collective)

    def _pipeline_applies(self, batch) -> bool:
        """The bucket pipeline needs the render to be the ONLY consumer of the effective weights in the step (the hooks hand slices of
        ITS gradient to the side stream): the fused schedule, auxiliary points riding in the render launches (tile-aligned sample count),
        no ray chunking, the fp32 chain.  Decided from the configuration and the batch SHAPE before the step, so that every rank of a
        job takes the same branch (= issues the same sequence of collectives)."""
        r, eng = self.renderer, self.renderer.engine
        if self.loss_fn is not compute_loss_fused and getattr(self.loss_fn, "func", None) is not compute_loss_fused:
            return False
        N = batch["rays"].shape[0]
        up = r.n_importance if r.n_importance > 0 else 0          # (important_begin_iter > step only makes the sample count smaller)
        for S in {r.n_samples, r.n_samples + up}:
            if N == 0 or (N * S) % 64:
                return False
        f = (_lib_flags_save(r))
        if r._chunk_rays(N, r.n_samples + up, f):
            return False
        return not (eng.split_precision and eng.x3_train_chain)

    def _train_step_pipelined(self, batch, global_step: int, u_perturb=None, u_neigh=None):
        from .parallel import allreduce_flat
        eng = self.renderer.engine
        if self._ar_stream is None:
            self._ar_stream = torch.cuda.Stream(device=eng.device)
        eng._grad_pipeline = dict(dflat=None, dweff_ptr=None, remaining=None, adopted=False)
        eng.wgrad_stage_hook = self._pipeline_hook
        try:
            loss, terms, ret = self._step_body(batch, global_step, u_perturb, u_neigh)
        finally:
            eng.wgrad_stage_hook = None
            pipe, eng._grad_pipeline = eng._grad_pipeline, None
            # whatever happened, nothing of this step may still be running on the side stream when its buffers (slices of the step
            # arena) are reused
            torch.cuda.current_stream(eng.device).wait_stream(self._ar_stream)
        if pipe["dflat"] is None or not pipe["adopted"]:
            # _pipeline_applies said the render is the only consumer of weff, yet the weight-norm backward did not receive the buffer the
            # hooks worked on: the ranks' collective sequences can no longer be trusted to match
            raise RuntimeError("overlap_allreduce: the staged backward and the weight-norm backward disagree about the gradient buffer "
                               "(a second consumer of the effective weights in this step?); run this configuration with overlap_allreduce=False")
        g = self.optimizer.flat_grad(include_variance=True)
        ev = self.allreduce_events
        if ev is not None:
            ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            ev[-1][0].record()
        world = 1
        for a, b in self._bucket_plan()["C"]:
            world = allreduce_flat(g[a:b], group=self.group, force=self.force_collective)
        if ev is not None:
            ev[-1][1].record()
        self.pipelined_steps += 1
        self.optimizer.step(grad=g, grad_scale=1.0 / world, variance_in_grad=True)
        return loss.detach(), terms, ret

    def _train_step(self, batch, global_step: int, u_perturb=None, u_neigh=None):
        if self.overlap_allreduce and self.data_parallel and isinstance(self.optimizer, FlatAdam) and self._pipeline_applies(batch):
            return self._train_step_pipelined(batch, global_step, u_perturb, u_neigh)
        loss, terms, ret = self._step_body(batch, global_step, u_perturb, u_neigh)
        if isinstance(self.optimizer, FlatAdam):
            if self.data_parallel:       # ONE all-reduce (sum) of the flat gradient bucket; the 1/world scale rides in the update
                from .parallel import allreduce_flat
                g = self.optimizer.flat_grad(include_variance=True)
                ev = self.allreduce_events
                if ev is not None:      # measurement: HIP events around the step's one data-path collective (bench.py collective_proof)
                    ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                    ev[-1][0].record()
                world = allreduce_flat(g, group=self.group, force=self.force_collective)
                if ev is not None:
                    ev[-1][1].record()
                self.optimizer.step(grad=g, grad_scale=1.0 / world, variance_in_grad=True)
            else:
                self.optimizer.step()
            return loss.detach(), terms, ret
        if self.data_parallel:
            from .parallel import allreduce_gradients
            allreduce_gradients(self.params, group=self.group)
        self.optimizer.step()
        return loss.detach(), terms, ret
