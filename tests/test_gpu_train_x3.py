"""The OPT-IN split-precision TRAINING chain (csrc/infer_x3r.hip with saves + csrc/train_x3r.hip): the forward writes the saved stacks
the backward and the weight-gradient GEMMs read in the SAME workspace buffers and layouts as the fp32 kernels, so both families are
compared buffer by buffer on the same points, and the parameter gradient of the whole chain is compared with the fp32 chain's and with
autograd on the fp64 oracle (budgets of tests/test_gpu_backward.py)."""
import numpy as np
import pytest
import torch

import test_gpu_backward as B
from oracle_util import CASES
from test_gpu_point import _setup, qd

pytestmark = pytest.mark.gpu

# WsBuf ids (csrc/workspace.h) of the saved stacks compared below
WS = dict(XC=0, V=1, SDF=2, FEAT=3, GC=4, GO=5, RGB=6, S_ACT=7, D_U0=8, D_U=9, D_MASK=10, D_R=11, S_S0=12, S_RHO=13, S_ADJEPS=14, C_IN=15, C_H=16,
          C_MASK=17, C_Y=18, C_Y8=19, FEATBAR=20, XCBAR_C=21, GCBAR_C=22, VBAR_C=23, S_TAU0=24, S_TAU=25, S_ZB=26, XCBAR=27, JU=28, D_T0=29,
          D_T=30, D_A=31, D_A8=32, C_SBAR=33)


def _buf(ctx, name, rows, width, layers=1):
    """[layers][rows][width] view of a workspace buffer (row-major stacks only)."""
    eng = ctx.eng
    off = int(eng.lib.es_point_workspace_offset(ctx.M, ctx.flags, WS[name]))
    return ctx.ws[off:off + layers * rows * width].view(layers, rows, width)


def _stack(ctx, name, frag):
    """[8][Mp][256] view of one of the SDF network's stacks: row-major in the split-precision family, accumulator-fragment order
    (csrc/chain_common.h frag_off: [64-row tile][w][ri][ni][q][hi][lo] x 4 rows) in the fp32 family."""
    Mp = ctx.Mp
    a = _buf(ctx, name, Mp, 256, 8)
    if not frag:
        return a
    t = a.view(8, Mp // 64, 4, 2, 2, 4, 2, 32, 4)              # L, tile, w, ri, ni, q, hi, lo, i
    return t.permute(0, 1, 3, 5, 6, 8, 2, 4, 7).reshape(8, Mp, 256)      # rows (tile, ri, q, hi, i), columns (w, ni, lo)


def _points(M, seed):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.uniform(-0.8, 0.8, size=(M, 3)).astype(np.float32)).cuda()
    d = rng.normal(size=(M, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return x, torch.from_numpy(rng.uniform(size=(M,)).astype(np.float32)).cuda(), torch.from_numpy(d.astype(np.float32)).cuda(), rng


def _close(a, b, q_tol, q=0.999, frac_bad=2e-3):
    """Stacks behind ReLU masks: equal to rounding almost everywhere; a pre-activation within rounding of 0 may flip in one family and
    then changes a few downstream elements by O(1) -- bounded in number."""
    d = (a.double() - b.double()).abs()
    scale = float(b.double().abs().max()) + 1e-30
    assert float(torch.quantile(d.flatten()[:4_000_000], q)) <= q_tol * scale, (float(torch.quantile(d.flatten()[:4_000_000], q)), scale)
    assert float((d > 1e-3 * scale).double().mean()) <= frac_bad, float((d > 1e-3 * scale).double().mean())


@pytest.mark.parametrize("mode", ["init", "trained"])
@pytest.mark.parametrize("M,color", [(777, True), (4096, False), (20000, True)])
def test_forward_with_saves_matches_fp32_buffers(mode, M, color):
    """The family's deformation and colour kernels around the fp32 SDF kernels (the SDF network stays fp32 in the training chain)."""
    from endosurf_amd import _lib
    eng, flat, weff, packed, net = _setup(31, mode, True)
    x, t, d, _ = _points(M, M)
    flags = _lib.PF_DEFORM | (_lib.PF_COLOR if color else 0) | _lib.PF_SAVE
    eng.x3_infer_min = 1
    eng.split_precision = False
    ref = eng.point_forward(eng.points(x=x, t=t, dirs=d), weff, packed, flags)
    eng.split_precision = True
    ctx = eng.point_forward(eng.points(x=x, t=t, dirs=d), weff, packed, flags)
    torch.cuda.synchronize()
    assert ctx.x3_chain and not ref.x3_chain
    assert qd(ctx.view("xc"), ref.view("xc")) < 3e-6 and qd(ctx.view("sdf"), ref.view("sdf")) < 1e-5
    assert qd(ctx.view("v"), ref.view("v"), 0.99) < 5e-5 and qd(ctx.view("go"), ref.view("go"), 0.98) < 2e-4
    Mp = ctx.Mp
    for name, rows, width, layers in (("D_U0", 2 * Mp, 64, 1), ("D_U", 2 * Mp, 256, 8), ("D_R", Mp, 256, 8)):
        a, b = _buf(ctx, name, rows, width, layers), _buf(ref, name, rows, width, layers)
        real = 2 * M if rows == 2 * Mp else M
        cols = 52 if name == "D_U0" else width
        _close(a[:, :real, :cols], b[:, :real, :cols], 2e-6)
    # layer 3 of the deformation network has 204 outputs: its adjoint columns beyond are exact zeros in both families
    assert float(_buf(ctx, "D_R", Mp, 256, 8)[3, :M, 204:].abs().max()) == 0.0
    assert qd(ctx.view("gc"), ref.view("gc")) < 1e-4
    for name in ("S_S0", "S_ADJEPS"):
        _close(_buf(ctx, name, Mp, 64)[:, :M, :39], _buf(ref, name, Mp, 64)[:, :M, :39], 2e-5)
    for name in ("S_ACT", "S_RHO"):
        _close(_stack(ctx, name, True)[:, :M], _stack(ref, name, True)[:, :M], 2e-5)
    if color:
        assert qd(ctx.view("feat"), ref.view("feat")) < 5e-5
        assert qd(ctx.view("rgb"), ref.view("rgb"), 0.98) < 5e-5
        _close(_buf(ctx, "C_IN", Mp, 128)[:, :M, :93], _buf(ref, "C_IN", Mp, 128)[:, :M, :93], 5e-5, q=0.99, frac_bad=2e-2)      # incl. enc4(d_c): J d flips
        _close(_buf(ctx, "C_H", Mp, 256, 8)[:, :M], _buf(ref, "C_H", Mp, 256, 8)[:, :M], 2e-5, q=0.99, frac_bad=2e-2)


@pytest.mark.parametrize("mode,color", [("init", True), ("trained", True), ("trained", False)])
def test_backward_matches_fp32_chain_and_saved_adjoints(mode, color):
    """Same points, same output adjoints: the saved backward stacks and the gradient w.r.t. the effective weights of the two families."""
    from endosurf_amd import _lib
    M = 5000
    eng, flat, weff, packed, net = _setup(33, mode, True)
    eng.deterministic = True
    x, t, d, rng = _points(M, 5)
    g = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32)).cuda()
    d_sdf, d_go, d_rgb = g(M, 1), g(M, 3), g(M, 3)
    flags = _lib.PF_DEFORM | (_lib.PF_COLOR if color else 0) | _lib.PF_SAVE
    eng.x3_infer_min = 1
    out = {}
    for split in (False, True):
        eng.split_precision = split
        ctx = eng.point_forward(eng.points(x=x, t=t, dirs=d), weff, packed, flags)
        assert ctx.x3_chain == split
        dweff = eng.point_backward(ctx, weff, packed, d_sdf, d_go, d_rgb if color else None)
        torch.cuda.synchronize()
        out[split] = (ctx, dweff)
    (ref, dref), (ctx, dx3) = out[False], out[True]
    Mp = ctx.Mp
    assert qd(_buf(ctx, "JU", Mp, 3)[0, :M], _buf(ref, "JU", Mp, 3)[0, :M], 0.99) < 2e-4 * float(_buf(ref, "JU", Mp, 3).abs().max())
    for name, rows, width, layers in (("D_T0", Mp, 64, 1), ("D_T", Mp, 256, 8), ("D_A", 2 * Mp, 256, 8)):
        a, b = _buf(ctx, name, rows, width, layers), _buf(ref, name, rows, width, layers)
        real = 2 * M if rows == 2 * Mp else M
        cols = 52 if name == "D_T0" else width
        _close(a[:, :real, :cols], b[:, :real, :cols], 5e-6, frac_bad=5e-3)
    a8, b8 = _buf(ctx, "D_A8", 2 * Mp, 4)[0, :2 * M], _buf(ref, "D_A8", 2 * Mp, 4)[0, :2 * M]
    _close(a8, b8, 2e-5, q=0.99, frac_bad=2e-2)          # the J d rows are seeded by the colour network (d_c flips)
    _close(_buf(ctx, "S_TAU0", Mp, 64)[:, :M, :39], _buf(ref, "S_TAU0", Mp, 64)[:, :M, :39], 2e-5, q=0.99, frac_bad=2e-2)
    for name in ("S_TAU", "S_ZB"):
        _close(_stack(ctx, name, True)[:, :M], _stack(ref, name, True)[:, :M], 2e-5, q=0.99, frac_bad=2e-2)
    _close(_buf(ctx, "XCBAR", Mp, 3)[:, :M], _buf(ref, "XCBAR", Mp, 3)[:, :M], 2e-5, q=0.99, frac_bad=2e-2)
    if color:
        for name, width, layers in (("C_Y", 256, 8), ("C_Y8", 4, 1), ("FEATBAR", 256, 1), ("XCBAR_C", 3, 1), ("GCBAR_C", 3, 1), ("VBAR_C", 3, 1)):
            _close(_buf(ctx, name, Mp, width, layers)[:, :M], _buf(ref, name, Mp, width, layers)[:, :M], 2e-5, q=0.99, frac_bad=2e-2)
    # gradient w.r.t. the effective weights: per layer tensor relative L2 (isolated ReLU flips move single rows)
    lay = eng.lib
    import ctypes as C
    for net_id in range(3):
        for l in range(9):
            wo, bo = C.c_int64(), C.c_int64()
            _lib.check(lay.es_weff_layout(net_id, l, C.byref(wo), C.byref(bo)), "es_weff_layout")
            nxt = eng.n_weff
            a, b = dx3[wo.value:bo.value].double(), dref[wo.value:bo.value].double()
            if float(b.norm()) < 1e-12:
                continue
            # (a ReLU flip in the deformation network moves J d of that point, hence its d_c and its whole colour-network contribution)
            assert float((a - b).norm()) <= 1e-2 * float(b.norm()), (net_id, l, float((a - b).norm()), float(b.norm()))


@pytest.mark.parametrize("mode,use_deform,color", [("init", True, True), ("trained", True, True), ("trained", True, False), ("trained", False, True)])
def test_point_backward_on_the_split_chain_vs_fp64_oracle(mode, use_deform, color, monkeypatch):
    """tests/test_gpu_backward.py::test_point_backward (autograd on the fp64 oracle, same budgets) with every chain kernel of the
    split-precision family: the threshold below which small evaluations stay on the fp32 kernels is lowered to 1 point."""
    from endosurf_amd.engine import Engine
    monkeypatch.setenv("ES_SPLIT_BF16", "1")
    orig = Engine.__init__

    def init(self, device):
        orig(self, device)
        self.x3_infer_min = 1
    monkeypatch.setattr(Engine, "__init__", init)
    B.test_point_backward(mode, use_deform, color)


@pytest.mark.parametrize("name", CASES)
def test_training_loss_param_grads_on_the_split_chain(name, monkeypatch):
    from endosurf_amd.engine import Engine
    monkeypatch.setenv("ES_SPLIT_BF16", "1")
    orig = Engine.__init__

    def init(self, device):
        orig(self, device)
        self.x3_infer_min = 1
    monkeypatch.setattr(Engine, "__init__", init)
    B.test_training_loss_param_grads(name)
