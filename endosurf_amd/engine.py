"""Thin torch <-> C-ABI plumbing: allocates device buffers with torch, hands raw pointers to libendosurf_hip
on torch's current HIP stream.  No arithmetic happens here; every method is one or a few kernel launches."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import check, es_composite_args, es_points, ptr, stream_ptr


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    return t


class Engine:
    """Per-device handle (stateless apart from the loaded library)."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.EndoSurfHipError(
                f"endosurf_amd runs its hot path on an AMD GPU through HIP only (got device {device!r}); there is no CPU path")
        self.lib = _lib.load()
        with torch.cuda.device(self.device):
            check(self.lib.es_init(), "es_init")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_param = int(self.lib.es_param_floats())
        self.n_weff = int(self.lib.es_weff_floats())
        self.n_packed = int(self.lib.es_packed_floats())
        import os
        # Supported environment variables (README): ES_SPLIT_BF16, ES_DETERMINISTIC (here), ES_WORKSPACE_GB (renderer).  Everything else
        # below is an attribute only -- measurement code sets ``renderer.engine.<name>``, nothing reads development switches from the
        # environment any more.
        # ray marching evaluates its proposals in blocks of this many steps with early exit (0: one launch over all proposals)
        self.march_block = 32
        # deterministic mode: batch sums and weight gradients are reduced in a fixed order instead of with fp32 atomics
        # (bit-identical results from run to run; a few % slower).  Also settable per renderer: ``renderer.engine.deterministic = True``
        self.deterministic = os.environ.get("ES_DETERMINISTIC", "0") not in ("0", "", "false", "False")
        self._wg_scratch = None
        # opt-in split-precision mode (csrc/query_x3.hip, wgrad.hip): the large no-grad SDF queries (coarse samples, ray-marching
        # proposals, field extraction) and the weight-gradient GEMMs run on the bf16 matrix pipes with every fp32 operand split
        # exactly into three bf16 planes (six partial products, fp32 accumulation): fp32-class accuracy at ~2.7x the fp32 MFMA rate.
        # NOT the default; also settable per renderer through render_cfg["split_precision"] / ``renderer.engine.split_precision = True``
        self.split_precision = os.environ.get("ES_SPLIT_BF16", "0") not in ("0", "", "false", "False")
        self._x3 = None
        self.x3_query_min = 8193
        self.x3_infer_min = 16384      # points: below this a launch of 64/128-point tiles does not fill the chip
        # split-precision mode: grad-enabled evaluations run the split-precision TRAINING chain (infer_x3r.hip with saves +
        # train_x3r.hip); False keeps the fp32 chain kernels under the split-precision queries / weight gradients (round-2 behaviour)
        self.x3_train_chain = True
        # (the SDF network stays on the fp32 kernels in the training chain: csrc/infer_x3r.hip)
        # step arena (round 4): inside Trainer's step ``zeros`` hands out slices of ONE buffer that one memset cleared (arena_begin)
        self._arena, self._arena_off, self._arena_on = None, 0, False
        self._ones1 = None
        self._rng_calls = 0
        self._rng_step_calls = 0          # draws of the current step with a device-resident step counter (uniform)
        # set by a data-parallel Trainer with overlap_allreduce: called as hook(stage, dweff) behind each weight-gradient launch of a step
        self.wgrad_stage_hook = None
        self._grad_pipeline = None          # state of a pipelined data-parallel step (trainer.Trainer._train_step_pipelined)

    def st(self):
        """torch's current HIP stream ON THIS ENGINE'S DEVICE.  The library launches on the current HIP device, so the caller must
        be inside ``torch.cuda.device(engine.device)`` (the renderer's public methods and autograd's backward are)."""
        if torch.cuda.current_device() != self.device.index:
            raise _lib.EndoSurfHipError(
                f"engine for {self.device} used while the current device is cuda:{torch.cuda.current_device()}; "
                f"wrap the call in torch.cuda.device({self.device.index})")
        return stream_ptr(self.device)

    def wg_scratch(self):
        """Scratch of the deterministic weight-gradient reduction (allocated once, ~80 MB), or None in the default mode."""
        if not self.deterministic:
            return None
        if self._wg_scratch is None:
            self._wg_scratch = self.empty(int(self.lib.es_wgrad_scratch_floats()))
        return self._wg_scratch

    # ---- buffers ----------------------------------------------------------------------------------
    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, device=self.device, dtype=dtype)

    def zeros(self, *shape, dtype=torch.float32):
        if self._arena_on and dtype == torch.float32:
            shp = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
            n = 1
            for s in shp:
                n *= int(s)
            off = self._arena_off
            if 0 < n and off + n <= self._arena.numel():
                self._arena_off = off + (n + 63) // 64 * 64          # 256-byte aligned slices
                return self._arena[off:off + n].view(shp)
        return torch.zeros(*shape, device=self.device, dtype=dtype)

    # ---- step arena / step plumbing (csrc/step.hip) --------------------------------------------------
    def arena_begin(self, extra_floats: int = 0):
        """Start of a training step: every ``zeros`` until ``arena_end`` is a slice of one buffer cleared by ONE memset (the eikonal
        sums, the inv_s adjoint, the effective-weight and parameter gradient buffers, the marching scratch ...) instead of a fill
        launch each.  The slices are valid until the next ``arena_begin`` (the flat gradient of a step lives here: it is consumed by the
        optimiser within the step; ``.grad`` views read as zeros once the next step has begun)."""
        need = self.n_weff + self.n_param + 4096 + int(extra_floats) + 64 * 32
        need = (need + 4095) // 4096 * 4096          # (a 16-KB multiple: the runtime clears it with one fill kernel, not body + remainder)
        if self._arena is None or self._arena.numel() < need:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.EndoSurfHipError("the step arena must exist before a step is captured (run one eager step first)")
            self._arena = self.empty(need)
        check(self.lib.es_zero(ptr(self._arena), 4 * self._arena.numel(), self.st()), "es_zero")
        self._arena_off, self._arena_on = 0, True
        self._rng_step_calls = 0

    def arena_end(self):
        self._arena_on = False

    @property
    def ones1(self) -> torch.Tensor:
        """A persistent device [1] holding 1.0: the seed of a step's backward pass (``loss.backward(gradient=...)``: no fill launch, and
        the loss node recognises it by address and hands its adjoints on unscaled)."""
        if self._ones1 is None:
            self._ones1 = torch.ones(1, device=self.device)
        return self._ones1

    def uniform(self, n: int, step_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
        """n uniform draws in [0, 1) in ONE launch (Philox4x32-10 keyed by torch's seed).  Eager steps: subsequence = this engine's call
        index.  Graph-mode steps: subsequence = 2^63 + the device-resident step counter, so that replays draw new numbers."""
        out = self.empty(int(n))
        seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        if step_dev is not None:
            # graph-mode steps (the two eager warm-up calls and every replay) draw from subsequences 2^63 + (k << 40) + step counter, k = the
            # index of the call within its step (reset by arena_begin): a space disjoint from the eager steps' call index below, so that
            # mixing train_step and train_step_graph never repeats a stream, and two draws of one step never share one
            sub = (1 << 63) + (self._rng_step_calls << 40)
            self._rng_step_calls += 1
        else:
            sub = self._rng_calls
            self._rng_calls += 1
        check(self.lib.es_uniform(ptr(out), int(n), seed, sub, ptr(step_dev) if step_dev is not None else None, self.st()), "es_uniform")
        return out

    # ---- weights ----------------------------------------------------------------------------------
    def weightnorm_pack(self, flat_params: torch.Tensor, use_deform: bool):
        weff = self.empty(self.n_weff)
        packed = self.empty(self.n_packed)
        if not use_deform:
            check(self.lib.es_zero(ptr(weff), 4 * weff.numel(), self.st()), "es_zero")
        check(self.lib.es_weightnorm_pack(ptr(flat_params), ptr(weff), ptr(packed), int(use_deform), self.st()), "es_weightnorm_pack")
        return weff, packed

    def weightnorm_backward_layers(self, flat_params, dweff, dparams, first_layer: int, n_layers: int):
        """Layers [first_layer, first_layer + n_layers) of the 27 (9 * network + layer) into their slices of ``dparams``."""
        check(self.lib.es_weightnorm_backward_layers(ptr(flat_params), ptr(dweff), ptr(dparams), int(first_layer), int(n_layers), self.st()),
              "es_weightnorm_backward_layers")

    def weightnorm_backward(self, flat_params, dweff, use_deform: bool):
        dparams = self.zeros(self.n_param)
        check(self.lib.es_weightnorm_backward(ptr(flat_params), ptr(dweff), ptr(dparams), int(use_deform), self.st()),
              "es_weightnorm_backward")
        return dparams

    # ---- point sources ----------------------------------------------------------------------------
    @staticmethod
    def points(x=None, t=None, dirs=None, rays=None, z=None, n_per_ray=1, ldz=None, M=None) -> es_points:
        """mode 0: explicit (x, t[, dirs]); mode 1: ray samples (rays, z); mode 2: ray samples followed by explicit (x, t)."""
        p = es_points()
        if rays is not None:
            p.rays, p.z = ptr(rays), ptr(z)
            p.n_per_ray = int(n_per_ray)
            p.ldz = int(ldz if ldz is not None else z.shape[-1])
            n_ray_pts = rays.shape[0] * n_per_ray
            if x is not None:
                p.mode, p.M_split = 2, n_ray_pts
                p.x, p.t = ptr(x), ptr(t)
                p.t_scalar = 0
                p.M = n_ray_pts + x.shape[0]
            else:
                p.mode = 1
                p.M = int(M if M is not None else n_ray_pts)
            p._keep = (rays, z, x, t)
        else:
            p.mode = 0
            p.x, p.t = ptr(x), ptr(t)
            p.dirs = ptr(dirs) if dirs is not None else None
            p.t_scalar = 1 if t.numel() == 1 and x.shape[0] != 1 else 0
            p.n_per_ray, p.ldz = 1, 1
            p.M = int(M if M is not None else x.shape[0])
            p._keep = (x, t, dirs)
        return p

    def packed_x3(self, weff, use_deform: bool):
        """The split-precision packing of ``weff``, one entry per ``use_deform`` (a deform model also evaluates canonical-space points
        with the deformation network switched off), rebuilt when a new effective-weight buffer shows up.  An entry keeps its source
        buffer alive (so that its address cannot be recycled for other weights) and the event recorded behind the packing launch:
        a consumer on ANOTHER stream waits for it (the sampling chain of a training step runs on a side stream)."""
        if self._x3 is None or self._x3["ptr"] != weff.data_ptr():
            self._x3 = {"ptr": weff.data_ptr(), "src": weff.detach(), "packs": {}}
        packs = self._x3["packs"]
        cur = torch.cuda.current_stream(self.device)
        e = packs.get(bool(use_deform))
        if e is None:
            buf = torch.empty(int(self.lib.es_packed_x3_bytes()), device=self.device, dtype=torch.uint8)
            check(self.lib.es_pack_x3(ptr(weff), ptr(buf), int(use_deform), self.st()), "es_pack_x3")
            ev = torch.cuda.Event()
            ev.record(cur)
            e = packs[bool(use_deform)] = (buf, ev, cur)
        elif e[2] != cur:
            cur.wait_event(e[1])
            if not torch.cuda.is_current_stream_capturing():
                e[0].record_stream(cur)
        return e[0]

    def x3_buffers(self):
        """The split-precision packings currently cached (a captured hipGraph keeps them alive: it holds their raw pointers)."""
        return [] if self._x3 is None else [self._x3["src"]] + [e[0] for e in self._x3["packs"].values()]

    def _use_x3(self, M: int) -> bool:
        # small batches (the 1024-point secant queries, the 8 192-point up-sampling queries) are latency-bound chains: they stay on the
        # 16-point fp32 tiles.  The library's split-precision query runs 32-point LDS-resident tiles up to 8 192 points, measured SLOWER
        # there (0.27 vs 0.21 ms at 8 192 points, 0.27 vs 0.17 at 1 024: DESIGN 4, round 3), so the threshold stays above them
        return self.split_precision and M >= self.x3_query_min

    def query_sdf(self, pts: es_points, weff, packed, use_deform: bool, tile_points: int = 0) -> torch.Tensor:
        """``tile_points``: 16 / 32 / 64 points per workgroup, 0 = the library's choice by batch size (es_query_sdf_tiles)."""
        out = self.empty(pts.M)
        if self._use_x3(pts.M):
            check(self.lib.es_query_sdf_x3(C.byref(pts), ptr(self.packed_x3(weff, use_deform)), ptr(weff), ptr(out), 0, None, int(use_deform),
                                           self.st()), "es_query_sdf_x3")
            return out
        if tile_points:
            check(self.lib.es_query_sdf_tiles(C.byref(pts), ptr(packed), ptr(weff), ptr(out), int(use_deform), int(tile_points), self.st()),
                  "es_query_sdf_tiles")
            return out
        check(self.lib.es_query_sdf(C.byref(pts), ptr(packed), ptr(weff), ptr(out), int(use_deform), self.st()), "es_query_sdf")
        return out

    # ---- per-ray kernels --------------------------------------------------------------------------
    def ray_setup(self, rays, u, n, sample_dist, lin_mode, z, want_bounds=False):
        N = rays.shape[0]
        near = self.empty(N) if want_bounds else None
        far = self.empty(N) if want_bounds else None
        check(self.lib.es_ray_setup(ptr(rays), ptr(u) if u is not None else None, N, n, float(sample_dist), lin_mode, ptr(z),
                                    z.shape[1], ptr(near), ptr(far), self.st()), "es_ray_setup")
        return near, far

    def sample_z(self, rays, u_perturb, weff, packed, use_deform, n_samples, n_importance, up_sample_steps, upsample: bool,
                 trace: Optional[list] = None, racing: bool = False):
        """Coarse sampling + SDF-guided hierarchical up-sampling (reference render_rays, endosurf.py:71-110).
        Returns z [N, S] (S = n_samples (+ n_importance)).  ``racing``: this chain shares the GPU with another chain of small launches
        (the secant iterations of a training step): its coarse query runs on 32-point tiles (es_query_sdf_tiles)."""
        N = rays.shape[0]
        n = n_samples
        sample_dist = 2.0 / n_samples
        do_up = upsample and n_importance > 0 and up_sample_steps > 0
        S = n + (n_importance if do_up else 0)
        if do_up and not racing and trace is None and N > 0 and n_importance % up_sample_steps == 0 and not self._use_x3(N * n):
            # the same launches in the same order, issued by ONE library call (es_sample_z) instead of ~15: a step that ends in a host
            # sync (the reference trainer's loss.item()) starts with an empty queue, and this chain of small launches is where the GPU
            # would wait for the host
            z_out = self.empty(N, S)
            scratch = self.empty(int(self.lib.es_sample_scratch_floats(N, n_samples, n_importance, up_sample_steps)))
            check(self.lib.es_sample_z(ptr(rays), ptr(u_perturb) if u_perturb is not None else None, N, n_samples, n_importance, up_sample_steps, 1,
                                       ptr(packed), ptr(weff), int(use_deform), ptr(z_out), ptr(scratch), self.st()), "es_sample_z")
            return z_out
        zc = self.empty(N, S)
        self.ray_setup(rays, u_perturb, n, sample_dist, 0, zc)
        if trace is not None:
            trace.append(zc[:, :n].clone())
        if not do_up:
            return zc
        n_imp = n_importance // up_sample_steps
        zn = self.empty(N, S)
        # (racing: never TALLER than 32 points -- a batch the library would give 16- or 32-point tiles anyway keeps the library's choice)
        sdf_c = self.query_sdf(self.points(rays=rays, z=zc, n_per_ray=n, ldz=S), weff, packed, use_deform,
                               tile_points=_lib.QUERY_TILE_RACING if racing and N * n > 16384 else 0).view(N, n)
        ld_sdf = n
        sdf_a, sdf_b = self.empty(N, S), self.empty(N, S)
        src = self.empty(N, S, dtype=torch.int32)
        z_new = self.empty(N, n_imp)
        for i in range(up_sample_steps):
            check(self.lib.es_upsample_step(ptr(rays), ptr(zc), S, ptr(sdf_c), ld_sdf, N, n, n_imp, float(64 * 2 ** i), ptr(z_new),
                                            ptr(zn), S, ptr(src), self.st()), "es_upsample_step")
            if i + 1 != up_sample_steps:
                sdf_new = self.query_sdf(self.points(rays=rays, z=z_new, n_per_ray=n_imp, ldz=n_imp), weff, packed, use_deform)
                dst = sdf_a if sdf_c.data_ptr() != sdf_a.data_ptr() else sdf_b
                check(self.lib.es_merge_sdf(ptr(sdf_c), ld_sdf, ptr(sdf_new), n_imp, ptr(src), S, N, n, ptr(dst), self.st()), "es_merge_sdf")
                sdf_c, ld_sdf = dst, S
            zc, zn = zn, zc
            n += n_imp
            if trace is not None:
                trace.append(zc[:, :n].clone())
        return zc

    def mid_z(self, z, sample_dist):
        N, S = z.shape
        mid = self.empty(N, S)
        check(self.lib.es_mid_z(ptr(z), S, N, S, float(sample_dist), ptr(mid), self.st()), "es_mid_z")
        return mid

    def composite_args(self, rays, z, sdf, g_o, rgb, variance, sample_dist, cos_anneal) -> es_composite_args:
        a = es_composite_args()
        N, S = z.shape
        a.rays, a.z, a.ldz = ptr(rays), ptr(z), S
        a.sdf, a.g_o, a.rgb, a.variance = ptr(sdf), ptr(g_o), ptr(rgb), ptr(variance)
        a.N, a.S, a.sample_dist = N, S, float(sample_dist)
        if torch.is_tensor(cos_anneal):          # device scalar (a captured training step updates it between replays)
            a.cos_anneal, a.cos_anneal_dev = 0.0, ptr(cos_anneal)
        else:
            a.cos_anneal, a.cos_anneal_dev = float(cos_anneal), None
        a._keep = [rays, z, sdf, g_o, rgb, variance, cos_anneal]
        if self.deterministic:
            part = self.empty(N, 2)
            a.ray_part = ptr(part)
            a._keep.append(part)
        return a

    def composite_forward(self, a: es_composite_args, eik_acc=None, go_copy: bool = False):
        """``eik_acc`` [2] (optional): accumulate the eikonal sums into an existing buffer (chunked renders).  ``go_copy``: the kernel
        also writes the samples' g_o rows into storage of their own (out["go"] [N,S,3]: the renderer's ``gradients_o``)."""
        N, S = a.N, a.S
        out = dict(color=self.empty(N, 3), depth=self.empty(N, 1), weights=self.empty(N, S), cdf=self.empty(N, S),
                   weight_max=self.empty(N, 1), eik_acc=eik_acc if eik_acc is not None else self.zeros(2),
                   wmax_idx=self.empty(N, dtype=torch.int32))
        for k, v in out.items():
            setattr(a, k, ptr(v))
        if go_copy:
            out["go"] = self.empty(N, S, 3)
            a.go_copy = ptr(out["go"])
        a._keep.append(out)
        check(self.lib.es_composite_forward(C.byref(a), self.st()), "es_composite_forward")
        return out

    def composite_backward(self, a: es_composite_args, g_color, g_depth, g_eik, eik_den, g_weights=None, g_cdf=None, g_wmax=None,
                           g_gradients_o=None, d_invs_acc=None, n_aux: int = 0, g_aux_sdf=None, g_aux_go=None):
        """``n_aux`` > 0: d_sdf / d_go get n_aux extra rows behind the N*S sample rows, filled by the same launch with the auxiliary
        points' adjoints (None = zeros): the row order of the fused point evaluation, no concatenation."""
        N, S = a.N, a.S
        out = dict(d_sdf=self.empty(N * S + n_aux), d_go=self.empty(N * S + n_aux, 3), d_rgb=self.empty(N * S, 3),
                   d_invs_acc=d_invs_acc if d_invs_acc is not None else self.zeros(1))
        keep = [g_color, g_depth, g_eik, eik_den, g_weights, g_cdf, g_wmax, g_gradients_o, g_aux_sdf, g_aux_go]
        a.n_aux = int(n_aux)
        a.g_aux_sdf = ptr(g_aux_sdf) if (n_aux and g_aux_sdf is not None) else None
        a.g_aux_go = ptr(g_aux_go) if (n_aux and g_aux_go is not None) else None
        a.g_color, a.g_depth, a.g_eik, a.eik_den = ptr(g_color), ptr(g_depth), ptr(g_eik), ptr(eik_den)
        a.g_weights = ptr(g_weights) if g_weights is not None else None
        a.g_cdf = ptr(g_cdf) if g_cdf is not None else None
        a.g_wmax = ptr(g_wmax) if g_wmax is not None else None
        a.g_gradients_o = ptr(g_gradients_o) if g_gradients_o is not None else None
        for k, v in out.items():
            setattr(a, k, ptr(v))
        a._keep += keep + [out]
        check(self.lib.es_composite_backward(C.byref(a), self.st()), "es_composite_backward")
        return out

    # ---- ray marching ------------------------------------------------------------------------------
    def variance_terms(self, variance, d_invs_acc=None):
        """SingleVarianceNetwork's scalar epilogue: s_val = 1 / inv_s, or (with ``d_invs_acc``) the gradient of the variance."""
        out = self.empty(1)
        v = _f32(variance).reshape(1)
        if d_invs_acc is None:
            check(self.lib.es_variance_terms(ptr(v), None, ptr(out), None, self.st()), "es_variance_terms")
        else:
            check(self.lib.es_variance_terms(ptr(v), ptr(d_invs_acc), None, ptr(out), self.st()), "es_variance_terms")
        return out

    def march_begin(self, rays, weff, packed, use_deform, n_steps=128, tau=0.0):
        """First half of ray_marching (reference endosurf.py:352-406): SDF at n_steps proposals per ray (one big launch) and the
        first sign change -> secant bracket. Returns the state consumed by march_refine."""
        N = rays.shape[0]
        dprop = self.empty(N, n_steps)
        self.ray_setup(rays, None, n_steps, 0.0, 1, dprop)
        B = self.march_block
        if B and n_steps % B == 0 and n_steps > B and N * B >= 16384:
            # proposals in blocks of B steps; a ray is finished at its first sign change (nothing behind it can change the
            # reference's result), and tiles whose rays are all finished return at once
            sdf = self.zeros(N, n_steps)                 # skipped proposals read as 0: no sign change
            done = self.empty(N, dtype=torch.int32)
            for b in range(n_steps // B):
                p = self.points(rays=rays, z=dprop, n_per_ray=B, ldz=n_steps)
                p.z = C.c_void_p(dprop.data_ptr() + 4 * b * B)
                if self._use_x3(p.M):
                    check(self.lib.es_query_sdf_x3(C.byref(p), ptr(self.packed_x3(weff, use_deform)), ptr(weff),
                                                   C.c_void_p(sdf.data_ptr() + 4 * b * B), n_steps, ptr(done) if b else None, int(use_deform),
                                                   self.st()), "es_query_sdf_x3")
                else:
                    check(self.lib.es_query_sdf_rays(C.byref(p), ptr(packed), ptr(weff), C.c_void_p(sdf.data_ptr() + 4 * b * B), n_steps,
                                                     ptr(done) if b else None, int(use_deform), self.st()), "es_query_sdf_rays")
                if b + 1 < n_steps // B:
                    check(self.lib.es_march_progress(ptr(sdf), N, n_steps, (b + 1) * B, float(tau), ptr(done), self.st()), "es_march_progress")
        else:
            sdf = self.query_sdf(self.points(rays=rays, z=dprop, n_per_ray=n_steps, ldz=n_steps), weff, packed, use_deform)
        state = self.empty(N, 4)
        flags = self.empty(N, dtype=torch.int32)
        d_pred = self.empty(N)
        check(self.lib.es_march_find(ptr(sdf), ptr(dprop), N, n_steps, float(tau), ptr(state), ptr(flags), ptr(d_pred), self.st()),
              "es_march_find")
        return dict(rays=rays, weff=weff, packed=packed, use_deform=use_deform, tau=float(tau), state=state, flags=flags, d_pred=d_pred,
                    keep=(dprop, sdf))

    def march_refine(self, ms, n_secant_steps=8):
        """Second half (endosurf.py:410-449): n_secant_steps dependent secant iterations (latency-bound small launches)."""
        rays, N = ms["rays"], ms["rays"].shape[0]
        st = self.st()
        x = self.empty(N, 3)
        t = self.empty(N)
        for _ in range(n_secant_steps):
            check(self.lib.es_secant_points(ptr(rays), ptr(ms["d_pred"]), N, ptr(x), ptr(t), st), "es_secant_points")
            f_mid = self.query_sdf(self.points(x=x, t=t), ms["weff"], ms["packed"], ms["use_deform"])
            check(self.lib.es_secant_update(ptr(f_mid), N, ms["tau"], ptr(ms["state"]), ptr(ms["d_pred"]), st), "es_secant_update")
        d_out = self.empty(N, 1)
        check(self.lib.es_march_finish(ptr(ms["d_pred"]), ptr(ms["flags"]), N, ptr(d_out), st), "es_march_finish")
        return d_out

    def ray_marching(self, rays, weff, packed, use_deform, n_steps=128, n_secant_steps=8, tau=0.0):
        """ray_marching + secant (reference endosurf.py:344-449), fixed shape. Returns d_pred [N,1]."""
        N = rays.shape[0]
        B = self.march_block
        blocks = bool(B and n_steps % B == 0 and n_steps > B and N * B >= 16384)
        if N > 0 and not self._use_x3(N * (B if blocks else n_steps)):
            # ONE library call (es_ray_marching) issues the proposals' queries, the bracket search and the secant iterations (~30 launches)
            d_out = self.empty(N, 1)
            scratch = self.empty(int(self.lib.es_march_scratch_floats(N, n_steps)))
            check(self.lib.es_ray_marching(ptr(rays), N, int(n_steps), int(n_secant_steps), float(tau), int(B) if blocks else 0, ptr(packed), ptr(weff),
                                           int(use_deform), ptr(d_out), ptr(scratch), self.st()), "es_ray_marching")
            return d_out
        return self.march_refine(self.march_begin(rays, weff, packed, use_deform, n_steps, tau), n_secant_steps)


class PointCtx:
    """Workspace of one fused point evaluation + typed views of its outputs."""
    _OUT = {"xc": (_lib.WS_XC, 3), "v": (_lib.WS_V, 3), "sdf": (_lib.WS_SDF, 1), "feat": (_lib.WS_FEAT, 256),
            "gc": (_lib.WS_GC, 3), "go": (_lib.WS_GO, 3), "rgb": (_lib.WS_RGB, 3), "curv": (_lib.WS_CURV, 3), "xcbar": (_lib.WS_XCBAR, 3),
            "tbar": (_lib.WS_TBAR, 1), "vbar": (_lib.WS_VBAR, 3)}

    def __init__(self, eng: "Engine", pts, flags: int, m_color: int = 0):
        self.eng, self.pts, self.flags, self.M, self.m_color = eng, pts, flags, pts.M, int(m_color)
        self.x3_chain, self.px3 = False, None      # set by Engine.point_forward when the split-precision training chain produced it
        self.Mp = (self.M + 127) // 128 * 128        # csrc/workspace.h round_up_rows
        n = int(eng.lib.es_point_workspace_floats(self.M, flags))
        self.ws = eng.empty(max(n, 1))

    def view(self, name):
        buf, width = self._OUT[name]
        off = int(self.eng.lib.es_point_workspace_offset(self.M, self.flags, buf))
        v = self.ws[off:off + self.Mp * width].view(self.Mp, width)[:self.M]
        return v


def _point_forward(self, pts, weff, packed, flags: int, m_color: int = 0, fp32_only: bool = False) -> PointCtx:
    """``fp32_only``: keep the evaluation on the fp32 kernels in split-precision mode too (the point adjoint reads their mask words)."""
    ctx = PointCtx(self, pts, flags, m_color)
    save = bool(flags & _lib.PF_SAVE)
    if self.split_precision and pts.M >= self.x3_infer_min and (self.x3_train_chain or not save) and not fp32_only:
        # opt-in: the launches of a large evaluation in split precision -- csrc/infer_x3r.hip without PF_SAVE; with PF_SAVE the
        # split-precision TRAINING chain, whose workspace must go through es_point_backward_x3 (``ctx.x3_chain``)
        px3 = self.packed_x3(weff, bool(flags & _lib.PF_DEFORM))
        check(self.lib.es_point_forward_x3(C.byref(pts), ptr(packed), ptr(px3), ptr(weff), ptr(ctx.ws), flags, int(m_color),
                                           self.st()), "es_point_forward_x3")
        ctx.x3_chain = save
        ctx.px3 = px3
    else:
        check(self.lib.es_point_forward(C.byref(pts), ptr(packed), ptr(weff), ptr(ctx.ws), flags, int(m_color), self.st()), "es_point_forward")
    return ctx


Engine.point_forward = _point_forward


def _point_forward_rows(self, ctx: PointCtx, weff, packed, row0: int, nrows: int):
    """Rows [row0, row0 + nrows) of a workspace that is filled piece by piece (es_point_forward_rows): the colour part of a render whose
    workspace has room for the colour-less points of later calls, or one of those calls' pieces of the tail.  fp32 kernels."""
    check(self.lib.es_point_forward_rows(C.byref(ctx.pts), ptr(packed), ptr(weff), ptr(ctx.ws), ctx.flags, ctx.m_color, int(row0), int(nrows),
                                         self.st()), "es_point_forward_rows")


Engine.point_forward_rows = _point_forward_rows


def _point_backward(self, ctx: PointCtx, weff, packed, d_sdf, d_go, d_rgb=None, dweff=None, staged: bool = False):
    """Adjoints of (sdf [M,1], g_o [M,3], rgb [M,3]) -> gradient w.r.t. the effective-weight buffer (accumulated into
    ``dweff`` if given).  ``staged``: this call produces the WHOLE gradient of a step (not one chunk of several), so
    ``engine.wgrad_stage_hook(stage, dweff)`` -- if set -- may be called behind every network's weight-gradient launch."""
    M = ctx.M
    z = lambda g, w: (g.detach().to(torch.float32).contiguous() if g is not None else self.zeros(M, w))
    d_sdf, d_go = z(d_sdf, 1), z(d_go, 3)
    color = bool(ctx.flags & _lib.PF_COLOR)
    if color:
        mc = ctx.m_color if ctx.m_color > 0 else M
        d_rgb = d_rgb.detach().to(torch.float32).contiguous() if d_rgb is not None else self.zeros(mc, 3)
        assert d_rgb.shape[0] == mc
    if dweff is None:
        dweff = self.zeros(self.n_weff)
        if self._grad_pipeline is not None:          # a pipelined data-parallel step counts the gradient buffers of its weff consumers
            self._grad_pipeline["buffers"] = self._grad_pipeline.get("buffers", 0) + 1
    if ctx.x3_chain:      # the workspace of the split-precision training chain: that family's backward kernels
        check(self.lib.es_point_backward_x3(C.byref(ctx.pts), ptr(packed), ptr(ctx.px3), ptr(weff), ptr(ctx.ws), ctx.flags, ctx.m_color, ptr(d_sdf),
                                            ptr(d_go), ptr(d_rgb) if color else None, ptr(dweff), ptr(self.wg_scratch()), self.st()),
              "es_point_backward_x3")
        return dweff
    flags = ctx.flags | (_lib.PF_X3 if self.split_precision else 0)      # opt-in: weight-gradient GEMMs in split precision
    hook = self.wgrad_stage_hook if staged else None
    if hook is not None:
        # the same launches in four calls, with the caller's hook between them: a data-parallel trainer starts the all-reduce of a
        # network's gradient while the next network's weight-gradient launch runs (Trainer(overlap_allreduce=True))
        for stage in (_lib.BWD_CHAINS, _lib.BWD_WGRAD_DEFORM, _lib.BWD_WGRAD_SDF, _lib.BWD_WGRAD_COLOR):
            check(self.lib.es_point_backward_stages(C.byref(ctx.pts), ptr(packed), ptr(weff), ptr(ctx.ws), flags, ctx.m_color, ptr(d_sdf), ptr(d_go),
                                                    ptr(d_rgb) if color else None, ptr(dweff), ptr(self.wg_scratch()), stage, self.st()),
                  "es_point_backward_stages")
            if stage != _lib.BWD_CHAINS:
                hook(stage, dweff)
        return dweff
    check(self.lib.es_point_backward_det(C.byref(ctx.pts), ptr(packed), ptr(weff), ptr(ctx.ws), flags, ctx.m_color, ptr(d_sdf), ptr(d_go),
                                         ptr(d_rgb) if color else None, ptr(dweff), ptr(self.wg_scratch()), self.st()), "es_point_backward")
    return dweff


Engine.point_backward = _point_backward


def _point_input_adjoint(self, ctx: PointCtx, weff, packed, d_sdf, d_go):
    """The adjoint of the QUERY POINTS through g_o (and only through g_o: callers that expose sdf as a function of the points attach
    d sdf / d x = g_o themselves, renderer.EndoSurfNet.get_sdf_from_observed_space) -- the reference's create_graph=True second
    derivative (endosurf.py:581-601, :603-619).  Call right after point_backward on the same context:
        xbar = J^T xcbar - d_go * curv(g_c) - d_sdf * g_o        (include/endosurf_hip.h es_point_vjp; without a deformation network J = I)
    where xcbar (the backward's adjoint of x_c) carries the Hessian-vector product of the SDF network along J d_go, the curvature term is
    the deformation network's own second derivative, and the last term removes the sdf path's share of xcbar.  Overwrites the
    workspace's g_c / g_o / curvature buffers."""
    M = ctx.M
    if ctx.x3_chain:
        raise _lib.EndoSurfHipError("the point adjoint needs a workspace of the fp32 kernels (point_forward(..., fp32_only=True))")
    xcbar = ctx.view("xcbar")
    go = ctx.view("go").clone()
    if ctx.flags & _lib.PF_DEFORM:
        curv = ctx.view("curv").clone()
        ctx.view("gc").copy_(xcbar)
        check(self.lib.es_point_vjp(C.byref(ctx.pts), ptr(packed), ptr(weff), ptr(ctx.ws), ctx.flags, self.st()), "es_point_vjp")
        xbar = ctx.view("go").clone()
        if d_go is not None:
            xbar -= d_go.detach().to(torch.float32).reshape(M, 3) * curv
    else:
        xbar = xcbar.clone()
    if d_sdf is not None:
        xbar -= d_sdf.detach().to(torch.float32).reshape(M, 1) * go
    return xbar


Engine.point_input_adjoint = _point_input_adjoint



def _timing_enable(self, on: bool):
    self._timing_on = bool(on)          # (events cannot be recorded inside a captured graph: the renderer's captured forward stands down)
    check(self.lib.es_timing_enable(int(on)), "es_timing_enable")


def _timing_drain(self):
    """[(kernel name, rows, ms)] of every launch recorded since the last drain."""
    cap = 65536
    kid = (C.c_int * cap)()
    rows = (C.c_longlong * cap)()
    ms = (C.c_float * cap)()
    n = C.c_int()
    check(self.lib.es_timing_drain(cap, kid, rows, ms, C.byref(n)), "es_timing_drain")
    return [(self.lib.es_kernel_name(kid[i]).decode(), int(rows[i]), float(ms[i])) for i in range(n.value)]


Engine.timing_enable = _timing_enable
Engine.timing_drain = _timing_drain
