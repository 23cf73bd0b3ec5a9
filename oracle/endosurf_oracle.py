"""CPU oracle for EndoSurf's per-ray volume-rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``endosurf_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker / the timed CPU baseline.

It is an independent closed-form restatement (no ``autograd.grad`` in the
forward: forward-mode Jacobian for the deformation MLP, hand-written reverse
sweep for the SDF input gradient) of the algorithm in the reference's
``src/renderer`` (file:line cited per function, paths relative to the
reference root).  Parity is PINNED: ``tests/test_oracle_golden.py`` checks every
function here against vectors captured from the reference implementation itself
(``tools/make_golden.py`` imports the reference in the build container and
writes ``tests/golden/*.npz``).  Parameter gradients come from torch autograd
applied to this closed-form forward (double backward through ``sigmoid`` gives the
softplus'' terms) and are pinned against reference gradients the same way.

Everything is dtype-generic (float32 to mirror the reference, float64 to act as
a ground truth for tolerances) and runs on torch-CPU.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

SQRT2 = math.sqrt(2.0)

# Architecture of every config shipped by the reference
# (configs/endosurf/baseline/base_pull.yml:40-82; all 12 EndoSurf configs share
# it, only ``use_deform`` differs).
ARCH = dict(
    n_layers=9, hidden=256, skip=4,
    deform_in=52, deform_pos_L=6, deform_time_L=6,
    sdf_in=39, sdf_pos_L=6, sdf_out=257,
    color_in=349, color_pos_L=10, color_dir_L=4, feat=256,
)


# --------------------------------------------------------------------------------------
# encodings                                                        src/renderer/encoder.py
# --------------------------------------------------------------------------------------
def freq_encode(x: torch.Tensor, L: int) -> torch.Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] (encoder.py:40-54)."""
    out = [x]
    for i in range(L):
        f = float(2 ** i)
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def freq_encode_jac(x: torch.Tensor, L: int) -> torch.Tensor:
    """d freq_encode(x) / d x  ->  [M, D*(1+2L), D]  (closed form of the above)."""
    M, D = x.shape
    eye = torch.eye(D, dtype=x.dtype, device=x.device).expand(M, D, D)
    out = [eye]
    for i in range(L):
        f = float(2 ** i)
        out.append(eye * (f * torch.cos(x * f))[:, :, None])
        out.append(eye * (-f * torch.sin(x * f))[:, :, None])
    return torch.cat(out, 1)


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
def effective_weight(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """nn.utils.weight_norm (dim=0): w = g * v / ||v||_row  (utils.py:57-58, 108-109)."""
    return g * v / v.norm(dim=1, keepdim=True)


class OracleNet:
    """Holds a reference-format state dict and evaluates EndoSurfNet in closed form.

    ``params`` maps ``"{deform,sdf,color}_network.net.{l}.{bias,weight_g,weight_v}"`` and
    ``"deviation_network.variance"`` to tensors (SURVEY A.3 / endosurf.py:559-568).
    """

    def __init__(self, params: Dict[str, torch.Tensor], use_deform: bool = True):
        self.p = params
        self.use_deform = use_deform

    # -- helpers ---------------------------------------------------------------------
    def _wb(self, net: str, l: int):
        g = self.p[f"{net}.net.{l}.weight_g"]
        v = self.p[f"{net}.net.{l}.weight_v"]
        b = self.p[f"{net}.net.{l}.bias"]
        return effective_weight(g, v), b

    # -- deformation MLP -------------------------------------------------------------
    def deform(self, x: torch.Tensor, t: torch.Tensor, with_jac: bool = True):
        """DeformNetwork.forward (endosurf.py:724-738) + closed-form d(x+dx)/dx.

        Returns (delta_x [M,3], J [M,3,3] with J[m,i,k] = d x_c_i / d x_k), matching
        ``get_deform_grad_from_observed_space`` (endosurf.py:621-658).
        """
        M = x.shape[0]
        if not self.use_deform:
            eye = torch.eye(3, dtype=x.dtype).expand(M, 3, 3)
            return torch.zeros_like(x), eye
        if t.dim() == 1:
            t = t[:, None]
        e = torch.cat([freq_encode(x, 6), freq_encode(t, 6)], -1)          # [M,52]
        de = None
        if with_jac:
            de = torch.cat([freq_encode_jac(x, 6),
                            torch.zeros(M, 13, 3, dtype=x.dtype)], 1)      # [M,52,3]
        u, du = e, de
        a = da = None
        for l in range(9):
            W, b = self._wb("deform_network", l)
            if l == 4:                                                       # IDR skip
                u = torch.cat([u, e], -1) / SQRT2
                if with_jac:
                    du = torch.cat([du, de], 1) / SQRT2
            a = u @ W.t() + b
            if with_jac:
                da = torch.einsum("oi,mik->mok", W, du)
            if l != 8:
                m = (a > 0).to(a.dtype)
                u = a * m
                if with_jac:
                    du = da * m[:, :, None]
        J = None
        if with_jac:
            J = torch.eye(3, dtype=x.dtype)[None] + da
        return a, J

    # -- SDF MLP ---------------------------------------------------------------------
    @staticmethod
    def _softplus(z):
        """nn.Softplus(beta=100, threshold=20) (endosurf.py:771)."""
        return torch.where(z * 100.0 > 20.0, z, torch.log1p(torch.exp(torch.clamp(z * 100.0, max=20.0))) / 100.0)

    @staticmethod
    def _softplus_grad(z):
        return torch.where(z * 100.0 > 20.0, torch.ones_like(z), torch.sigmoid(z * 100.0))

    def sdf_net(self, x_c: torch.Tensor, with_grad: bool = True):
        """SDFNetwork.forward (endosurf.py:773-786) and, optionally, the analytic
        input gradient d sdf / d x_c (``get_sdf_grad_from_canonical_space``,
        endosurf.py:603-619) by a hand-written reverse sweep.

        Returns (sdf [M,1], feat [M,256], g_c [M,3] or None).
        """
        eps = freq_encode(x_c, 6)
        s = eps
        zs = []
        z = None
        for l in range(9):
            V, c = self._wb("sdf_network", l)
            if l == 4:                                                       # NeRF skip
                s = torch.cat([s, eps], -1) / SQRT2
            z = s @ V.t() + c
            zs.append(z)
            if l != 8:
                s = self._softplus(z)
        sdf, feat = z[:, :1], z[:, 1:]
        g_c = None
        if with_grad:
            V8, _ = self._wb("sdf_network", 8)
            adj_s = V8[0:1, :].expand(x_c.shape[0], -1)                      # d sdf / d s_8
            gamma = None
            adj_in = None
            for l in range(7, -1, -1):
                V, _ = self._wb("sdf_network", l)
                rho = self._softplus_grad(zs[l]) * adj_s                     # d sdf / d z_l
                adj_in = rho @ V                                             # d sdf / d (input of layer l)
                if l == 4:
                    gamma = adj_in[:, 256:] / SQRT2
                    adj_s = adj_in[:, :256] / SQRT2
                else:
                    adj_s = adj_in
            adj_eps = adj_in + gamma
            g_c = torch.einsum("mj,mjk->mk", adj_eps, freq_encode_jac(x_c, 6))
        return sdf, feat, g_c

    # -- colour MLP ------------------------------------------------------------------
    def color(self, x_c, g_c, d_c, feat):
        """ColorNetwork.forward (endosurf.py:828-842)."""
        inp = torch.cat([freq_encode(x_c, 10), g_c, freq_encode(d_c, 4), feat], -1)  # 349
        h = inp
        y = None
        for l in range(9):
            U, b = self._wb("color_network", l)
            if l == 4:
                h = torch.cat([h, inp], -1) / SQRT2
            y = h @ U.t() + b
            if l != 8:
                h = torch.relu(y)
        return torch.sigmoid(y)

    def inv_s(self):
        """SingleVarianceNetwork (endosurf.py:845-852) + clip (endosurf.py:168)."""
        return torch.exp(self.p["deviation_network.variance"] * 10.0).clamp(1e-6, 1e6)

    # -- composed queries ------------------------------------------------------------
    def sdf_observed(self, x, t):
        """get_sdf_from_observed_space (endosurf.py:570-579)."""
        dx, _ = self.deform(x, t, with_jac=False)
        sdf, _, _ = self.sdf_net(x + dx, with_grad=False)
        return sdf

    def point_eval(self, x, d, t, with_color: bool = True):
        """EndoSurfNet.forward (endosurf.py:660-689) + get_sdf_grad_from_observed_space
        (endosurf.py:581-601) using g_o = J^T g_c.

        Returns dict(sdf[M,1], rgb[M,3]|None, g_o[M,3], g_c, J, x_c, feat, d_c).
        """
        dx, J = self.deform(x, t, with_jac=True)
        x_c = x + dx
        sdf, feat, g_c = self.sdf_net(x_c, with_grad=True)
        g_o = torch.einsum("mik,mi->mk", J, g_c)
        rgb = d_c = None
        if with_color:
            v = torch.einsum("mik,mk->mi", J, d)
            d_c = v / (v.norm(dim=-1, keepdim=True) + 1e-10)
            rgb = self.color(x_c, g_c, d_c, feat)
        return dict(sdf=sdf, rgb=rgb, g_o=g_o, g_c=g_c, J=J, x_c=x_c, feat=feat, d_c=d_c)


# --------------------------------------------------------------------------------------
# ray utilities                                                     src/renderer/utils.py
# --------------------------------------------------------------------------------------
def sphere_intersection(o, d):
    """get_sphere_intersection (utils.py:194-210), r = 1. Returns near, far [N,1]."""
    d1 = -(d * o).sum(-1) / (d * d).sum(-1)
    p = o + d1[:, None] * d
    tmp = 1.0 - (p * p).sum(-1)
    d2 = torch.sqrt(torch.clamp(tmp, min=0.0)) / d.norm(dim=-1)
    near = torch.clamp(d1 - d2, min=0.0)
    far = d1 + d2
    return near[:, None], far[:, None]


def sample_pdf_det(bins, weights, n_samples):
    """sample_pdf(..., det=True) (utils.py:160-191)."""
    weights = weights + 1e-5
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, n_samples, dtype=torch.float32).to(bins.dtype)
    u = u.expand(cdf.shape[0], n_samples).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return bin_b + (u - cdf_b) / denom * (bin_a - bin_b)


def d_over_z(d):
    """rays_d / (rays_d.z + 1e-6) (endosurf.py:66)."""
    return d / (d[:, 2:] + 1e-6)


# --------------------------------------------------------------------------------------
# renderer                                                       src/renderer/endosurf.py
# --------------------------------------------------------------------------------------
class OracleRenderer:
    """Closed-form restatement of EndoSurfRenderer (endosurf.py:14-521)."""

    def __init__(self, net: OracleNet, render_cfg: dict):
        self.net = net
        self.anneal_end = render_cfg["anneal_end"]
        self.n_samples = render_cfg["n_samples"]
        self.perturb = render_cfg["perturb"]
        self.n_importance = render_cfg["n_importance"]
        self.important_begin_iter = render_cfg["important_begin_iter"]
        self.up_sample_steps = render_cfg["up_sample_steps"]

    def cos_anneal_ratio(self, iter_step):
        """get_cos_anneal_ratio (endosurf.py:215-219)."""
        if self.anneal_end == 0.0:
            return 1.0
        return min(1.0, iter_step / self.anneal_end)

    def up_sample(self, o, d, z, sdf, n_importance, inv_s):
        """up_sample (endosurf.py:221-266)."""
        dz = d_over_z(d)
        pts = o[:, None, :] + dz[:, None, :] * z[:, :, None]
        radius = pts.norm(dim=-1)
        inside = (radius[:, :-1] < 1.0) | (radius[:, 1:] < 1.0)
        prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
        prev_z, next_z = z[:, :-1], z[:, 1:]
        mid_sdf = (prev_sdf + next_sdf) * 0.5
        cos_val = (next_sdf - prev_sdf) / (next_z - prev_z + 1e-6)
        prev_cos = torch.cat([torch.zeros_like(cos_val[:, :1]), cos_val[:, :-1]], -1)
        cos_val = torch.minimum(prev_cos, cos_val).clamp(-1e3, 0.0) * inside.to(z.dtype)
        dist = next_z - prev_z
        prev_e = mid_sdf - cos_val * dist * 0.5
        next_e = mid_sdf + cos_val * dist * 0.5
        prev_cdf = torch.sigmoid(prev_e * inv_s)
        next_cdf = torch.sigmoid(next_e * inv_s)
        alpha = (prev_cdf - next_cdf + 1e-6) / (prev_cdf + 1e-6)
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
        return sample_pdf_det(z, alpha * T, n_importance)

    def cat_z_vals(self, o, d, time, z, new_z, sdf, last):
        """cat_z_vals (endosurf.py:268-287)."""
        N = z.shape[0]
        zc, index = torch.sort(torch.cat([z, new_z], -1), dim=-1)
        if not last:
            dz = d_over_z(d)
            pts = (o[:, None, :] + dz[:, None, :] * new_z[:, :, None]).reshape(-1, 3)
            t = time[:, None, None].expand(N, new_z.shape[1], 1).reshape(-1, 1)
            new_sdf = self.net.sdf_observed(pts, t).reshape(N, -1)
            sdf = torch.gather(torch.cat([sdf, new_sdf], -1), 1, index)
        return zc, sdf

    def sample_z(self, rays, iter_step, u_perturb: Optional[torch.Tensor]):
        """Coarse sampling + hierarchical up-sampling of render_rays (endosurf.py:63-110).

        ``u_perturb`` is the [N,1] uniform draw of endosurf.py:81 (None = no perturb).
        """
        o, d, time = rays[:, :3], rays[:, 3:6], rays[:, 8]
        N = rays.shape[0]
        near, far = sphere_intersection(o, d)
        sample_dist = 2.0 / self.n_samples
        t_vals = torch.linspace(0.0, 1.0, self.n_samples, dtype=torch.float32).to(rays.dtype)
        z = near + (far - near) * t_vals[None, :]
        if u_perturb is not None:
            z = z + (u_perturb - 0.5) * sample_dist
        trace = [z]
        if iter_step >= self.important_begin_iter and self.n_importance > 0:
            with torch.no_grad():
                dz = d_over_z(d)
                pts = (o[:, None, :] + dz[:, None, :] * z[:, :, None]).reshape(-1, 3)
                t = time[:, None, None].expand(N, self.n_samples, 1).reshape(-1, 1)
                sdf = self.net.sdf_observed(pts, t).reshape(N, self.n_samples)
                for i in range(self.up_sample_steps):
                    new_z = self.up_sample(o, d, z, sdf, self.n_importance // self.up_sample_steps, 64 * 2 ** i)
                    z, sdf = self.cat_z_vals(o, d, time, z, new_z, sdf, last=(i + 1 == self.up_sample_steps))
                    trace.append(z)
        return z, sample_dist, trace

    def render_core(self, o, d, time, z, sample_dist, cos_anneal_ratio):
        """render_core (endosurf.py:134-213)."""
        N, S = z.shape
        dz = d_over_z(d)
        dists = z[:, 1:] - z[:, :-1]
        dists = torch.cat([dists, torch.full_like(dists[:, :1], sample_dist)], -1)
        mid = z + dists * 0.5
        pts = (o[:, None, :] + dz[:, None, :] * mid[:, :, None]).reshape(-1, 3)
        dirs = d[:, None, :].expand(N, S, 3).reshape(-1, 3)
        t = time[:, None, None].expand(N, S, 1).reshape(-1, 1)
        pe = self.net.point_eval(pts, dirs, t, with_color=True)
        return self.composite(o, d, z, sample_dist, cos_anneal_ratio, pe["sdf"], pe["rgb"], pe["g_o"], self.net.inv_s())

    def composite(self, o, d, z, sample_dist, cos_anneal_ratio, sdf, rgb, g_o, inv_s):
        """Alpha compositing half of render_core (endosurf.py:168-213) from per-sample sdf [P,1], rgb [P,3], g_o [P,3]."""
        N, S = z.shape
        dz = d_over_z(d)
        dists = z[:, 1:] - z[:, :-1]
        dists = torch.cat([dists, torch.full_like(dists[:, :1], sample_dist)], -1)
        mid = z + dists * 0.5
        pts = (o[:, None, :] + dz[:, None, :] * mid[:, :, None]).reshape(-1, 3)
        dirs = d[:, None, :].expand(N, S, 3).reshape(-1, 3)
        sdf = sdf.reshape(-1, 1)
        rgb = rgb.reshape(N, S, 3)
        g_o = g_o.reshape(-1, 3)
        true_cos = (dirs * g_o).sum(-1, keepdim=True)
        r = cos_anneal_ratio
        iter_cos = -(torch.relu(-true_cos * 0.5 + 0.5) * (1.0 - r) + torch.relu(-true_cos) * r)
        dd = dists.reshape(-1, 1)
        next_sdf = sdf + iter_cos * dd * 0.5
        prev_sdf = sdf - iter_cos * dd * 0.5
        prev_cdf = torch.sigmoid(prev_sdf * inv_s)
        next_cdf = torch.sigmoid(next_sdf * inv_s)
        p, c = prev_cdf - next_cdf, prev_cdf
        alpha = ((p + 1e-6) / (c + 1e-6)).reshape(N, S).clamp(0.0, 1.0)
        relax = (pts.norm(dim=-1).reshape(N, S) < 1.2).to(z.dtype).detach()
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
        w = alpha * T
        depth = (w * mid).sum(-1, keepdim=True)
        color = (rgb * w[:, :, None]).sum(1)
        g3 = g_o.reshape(N, S, 3)
        eik = (g3.norm(dim=-1) - 1.0) ** 2
        eik = (relax * eik).sum() / (relax.sum() + 1e-6)
        return dict(color_map=color, depth_map=depth, gradients_o=g3, gradient_o_error=eik,
                    cdf=c.reshape(N, S), weights=w, s_val=(1.0 / inv_s), sdf=sdf.reshape(N, S),
                    rgb=rgb, mid_z=mid)

    def render_rays(self, rays, iter_step=0, u_perturb=None):
        """render_rays / forward (endosurf.py:54-132)."""
        o, d, time = rays[:, :3], rays[:, 3:6], rays[:, 8]
        N = rays.shape[0]
        z, sample_dist, trace = self.sample_z(rays, iter_step, u_perturb)
        ret = self.render_core(o, d, time, z, sample_dist, self.cos_anneal_ratio(iter_step))
        return dict(color_map=ret["color_map"], depth_map=ret["depth_map"], gradients_o=ret["gradients_o"],
                    gradient_o_error=ret["gradient_o_error"], weights=ret["weights"],
                    weight_max=ret["weights"].max(-1, keepdim=True)[0], cdf=ret["cdf"],
                    s_val=ret["s_val"].expand(N, z.shape[1]).mean(-1, keepdim=True),
                    z_vals=z, z_trace=trace, sdf=ret["sdf"], rgb=ret["rgb"], mid_z=ret["mid_z"])

    def errorondepth(self, rays, d_gt, mask):
        """errorondepth (endosurf.py:289-317). The relu(cos) sum is NOT masked (quirk kept)."""
        o, d, time = rays[:, :3], rays[:, 3:6], rays[:, 8]
        pts = o + d_over_z(d) * d_gt
        pe = self.net.point_eval(pts, d, time[:, None], with_color=False)
        relu_cos = torch.relu((d * pe["g_o"]).sum(-1, keepdim=True))
        inside = (pts.detach().norm(dim=-1, keepdim=True) < 1.0).to(rays.dtype) * mask
        denom = inside.sum() + 1e-6
        sdf_error = (inside * pe["sdf"]).abs().sum() / denom
        angle_error = relu_cos.abs().sum() / denom
        return sdf_error, angle_error, inside

    # -- offline helpers (surface rendering / meshing field), SURVEY 8f-3 ---------------------------------
    def renderonpts(self, pts, dirs, ts):
        """renderonpts (endosurf.py:502-521): colour and unit normal at given points; ``ts`` is [M,1] or the shared-time
        form [1] (DeformNetwork.forward, endosurf.py:726-727)."""
        sh = list(pts.shape[:-1])
        x, d = pts.reshape(-1, 3), dirs.reshape(-1, 3)
        t = ts[None, :].expand(x.shape[0], 1) if ts.dim() == 1 else ts.reshape(-1, 1)
        pe = self.net.point_eval(x, d, t, with_color=True)
        normal = pe["g_o"] / (pe["g_o"].norm(dim=-1, keepdim=True) + 1e-10)
        return pe["rgb"].reshape(*sh, 3), normal.reshape(*sh, 3)

    def renderondepth(self, rays, depth):
        """renderondepth (endosurf.py:450-488): colour / g_o at o + d_z*depth for rays with 0 < depth < inf, zeros elsewhere;
        d_out = depth with +inf replaced by the far sphere intersection."""
        o, d, time = rays[:, :3], rays[:, 3:6], rays[:, 8]
        _, far = sphere_intersection(o, d)
        valid = (depth[:, 0] > 0) & (depth[:, 0] != float("inf"))
        d_out = torch.where(depth == float("inf"), far, depth)
        color = torch.zeros(rays.shape[0], 3, dtype=rays.dtype)
        grad = torch.zeros(rays.shape[0], 3, dtype=rays.dtype)
        if valid.any():
            pts = o[valid] + d_over_z(d)[valid] * depth[valid]
            pe = self.net.point_eval(pts, d[valid], time[valid][:, None], with_color=True)
            color[valid] = pe["rgb"]
            grad[valid] = pe["g_o"]
        return color, grad, d_out

    def extract_fields(self, bound_min, bound_max, resolution, t):
        """extract_fields (utils.py:139-157) with the query of extract_observation_geometry (endosurf.py:490-500):
        u[i,j,k] = sdf(x_i, y_j, z_k; t) on linspace grids (the reference's 128-blocking only bounds memory)."""
        dt = bound_min.dtype
        ax = [torch.linspace(float(bound_min[i]), float(bound_max[i]), resolution, dtype=dt) for i in range(3)]
        xx, yy, zz = torch.meshgrid(*ax, indexing="ij")
        pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
        tt = torch.as_tensor(t, dtype=dt).reshape(1, 1).expand(pts.shape[0], 1)
        return self.net.sdf_observed(pts, tt).reshape(resolution, resolution, resolution)

    def ray_marching(self, rays, n_steps=128, n_secant_steps=8, tau=0.0):
        """ray_marching + secant (endosurf.py:344-449); n_steps is always 128 there
        (torch.randint(128, 129)). Returns d_pred [N,1] (inf = no hit, 0 = first point occupied)."""
        N = rays.shape[0]
        o, d, time = rays[:, :3], rays[:, 3:6], rays[:, 8:9]
        near, far = sphere_intersection(o, d)
        t_vals = torch.linspace(0.0, 1.0, n_steps, dtype=torch.float32).to(rays.dtype)
        d_prop = near * (1.0 - t_vals) + far * t_vals
        dz = d_over_z(d)
        pts = (o[:, None, :] + d_prop[:, :, None] * dz[:, None, :]).reshape(-1, 3)
        t = time[:, None, :].expand(N, n_steps, 1).reshape(-1, 1)
        with torch.no_grad():
            val = -(self.net.sdf_observed(pts, t).reshape(N, n_steps) - tau)
        mask_0_not_occupied = val[:, 0] < 0
        sign = torch.cat([torch.sign(val[:, :-1] * val[:, 1:]), torch.ones(N, 1, dtype=val.dtype)], -1)
        cost = sign * torch.arange(n_steps, 0, -1, dtype=val.dtype)
        values, indices = torch.min(cost, -1)
        ar = torch.arange(N)
        mask_sign_change = values < 0
        mask_neg_to_pos = val[ar, indices] < 0
        mask = mask_sign_change & mask_neg_to_pos & mask_0_not_occupied
        d_low = d_prop[ar, indices][mask]
        f_low = val[ar, indices][mask]
        ind2 = torch.clamp(indices + 1, max=n_steps - 1)
        d_high = d_prop[ar, ind2][mask]
        f_high = val[ar, ind2][mask]
        out = torch.ones(N, dtype=rays.dtype)
        if int(mask.sum()) != 0:
            rm = rays[mask]
            om, dm, tm = rm[:, :3], rm[:, 3:6], rm[:, 8:9]
            dzm = dm / dm[:, 2:]                                   # no epsilon here (endosurf.py:427)
            d_pred = -f_low * (d_high - d_low) / (f_high - f_low) + d_low
            for _ in range(n_secant_steps):
                p_mid = om + d_pred[:, None] * dzm
                with torch.no_grad():
                    f_mid = self.net.sdf_observed(p_mid, tm)[:, 0] - tau
                lo = f_mid < 0
                d_low = torch.where(lo, d_pred, d_low)
                f_low = torch.where(lo, f_mid, f_low)
                d_high = torch.where(lo, d_high, d_pred)
                f_high = torch.where(lo, f_high, f_mid)
                d_pred = -f_low * (d_high - d_low) / (f_high - f_low) + d_low
            out[mask] = d_pred
        out[~mask] = float("inf")
        out[~mask_0_not_occupied] = 0.0
        return out[:, None]

    def surface_neighbour_error(self, rays, mask, neighbour_rad, u_neigh: torch.Tensor):
        """surface_neighbour_error (endosurf.py:319-342). ``u_neigh`` [N,3] holds uniform draws;
        rows of valid rays are used (the reference draws rand_like(p_surf) for valid rays only)."""
        o, d, time = rays[:, :3], rays[:, 3:6], rays[:, 8]
        dz = d_over_z(d)
        with torch.no_grad():
            d_i = self.ray_marching(rays)
        valid = (torch.isfinite(d_i) & (d_i != 0) & (mask == 1))[:, 0]
        if not bool(valid.any()):
            return torch.zeros((), dtype=rays.dtype), d_i, valid
        p_surf = o[valid] + d_i[valid] * dz[valid]
        p_neig = p_surf + (u_neigh[valid] - 0.5) * neighbour_rad
        n = p_surf.shape[0]
        pp = torch.cat([p_surf, p_neig], 0)
        tt = torch.cat([time[valid], time[valid]], 0)[:, None]
        g = self.net.point_eval(pp, torch.zeros_like(pp), tt, with_color=False)["g_o"]
        normal = g / (g.norm(dim=-1, keepdim=True) + 1e-10)
        return (normal[:n] - normal[n:]).abs().mean(), d_i, valid


# --------------------------------------------------------------------------------------
# training-step loss                                    src/trainer/trainer_endosurf.py
# --------------------------------------------------------------------------------------
LOSS_WEIGHTS = dict(color=1.0, depth=1.0, sdf=1.0, angle=0.1, eikonal=0.1, surf_neig=0.1)


def train_loss(R: OracleRenderer, batch: Dict[str, torch.Tensor], iter_step: int,
               u_perturb, u_neigh, weights=LOSS_WEIGHTS, surf_neig_rad=0.1):
    """compute_loss (trainer_endosurf.py:106-162) without the logging."""
    rays, color_gt, depth_gt = batch["rays"], batch["color"], batch["depth"]
    mask_gt, cmask = batch["mask"], batch["color_mask"]
    ret = R.render_rays(rays, iter_step, u_perturb)
    color_loss = ((ret["color_map"] - color_gt) * cmask).abs().sum() / (cmask.sum() + 1e-10)
    sdf_loss, angle_loss, valid = R.errorondepth(rays, depth_gt, mask_gt)
    depth_loss = ((ret["depth_map"] - depth_gt) * valid * mask_gt).abs().sum() / ((valid * mask_gt).sum() + 1e-10)
    eik = ret["gradient_o_error"]
    sn, _, _ = R.surface_neighbour_error(rays, mask_gt, surf_neig_rad, u_neigh)
    total = (color_loss * weights["color"] + depth_loss * weights["depth"] + sdf_loss * weights["sdf"]
             + angle_loss * weights["angle"] + eik * weights["eikonal"] + weights["surf_neig"] * sn)
    terms = dict(color=color_loss, depth=depth_loss, sdf=sdf_loss, angle=angle_loss, eikonal=eik, surf_neig=sn)
    return total, terms, ret


def cal_psnr(a, b, mask):
    """cal_psnr (src/trainer/utils.py:340-353)."""
    import numpy as np
    a, b, mask = [t.detach().cpu().numpy() if torch.is_tensor(t) else t for t in (a, b, mask)]
    if mask.ndim == a.ndim - 1:
        mask = mask[..., None]
    mask_sum = np.sum(mask) + 1e-10
    return 20.0 * np.log10(1.0 / (((a - b) ** 2 * mask).sum() / (mask_sum * 3.0)) ** 0.5)


def lr_factor(it, n_iter=100000, warm_up_end=5000, alpha=0.05):
    """update_learning_rate (trainer_endosurf.py:183-203)."""
    if it < warm_up_end:
        return it / warm_up_end
    prog = (it - warm_up_end) / (n_iter - warm_up_end)
    return (math.cos(math.pi * prog) + 1.0) * 0.5 * (1 - alpha) + alpha
