"""GPU parity of the fused per-point network kernels (deform fwd+tangents, SDF fwd+reverse, colour) vs the fp64 oracle."""
import numpy as np
import pytest
import torch

import weightgen
from oracle import endosurf_oracle as O
from oracle_util import CASES, T, load_case, oracle_for

pytestmark = pytest.mark.gpu


def _setup(seed, mode, use_deform):
    from endosurf_amd import params
    from endosurf_amd.engine import Engine
    eng = Engine("cuda")
    state = weightgen.make_state(seed, mode, use_deform)
    flat = torch.from_numpy(params.flatten_state(state)).cuda()
    weff, packed = eng.weightnorm_pack(flat, use_deform)
    net = O.OracleNet({k: torch.tensor(v, dtype=torch.float64) for k, v in state.items()}, use_deform)
    return eng, flat, weff, packed, net


def qd(a, b, q=1.0):
    d = np.abs(a.detach().cpu().numpy().astype(np.float64) - b.detach().cpu().numpy().astype(np.float64))
    return float(d.max() if q >= 1.0 else np.quantile(d, q))


@pytest.mark.parametrize("mode,use_deform", [("init", True), ("trained", True), ("trained", False)])
@pytest.mark.parametrize("M,color", [(1, True), (100, True), (64, False), (777, True)])
def test_point_forward(mode, use_deform, M, color):
    from endosurf_amd import _lib
    eng, flat, weff, packed, net = _setup(31, mode, use_deform)
    rng = np.random.default_rng(M)
    x = torch.from_numpy(rng.uniform(-0.8, 0.8, size=(M, 3)).astype(np.float32))
    d = rng.normal(size=(M, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = torch.from_numpy(d.astype(np.float32))
    t = torch.from_numpy(rng.uniform(size=(M,)).astype(np.float32))
    flags = (_lib.PF_DEFORM if use_deform else 0) | (_lib.PF_COLOR if color else 0) | _lib.PF_SAVE
    ctx = eng.point_forward(eng.points(x=x.cuda(), t=t.cuda(), dirs=d.cuda()), weff, packed, flags)
    torch.cuda.synchronize()
    with torch.no_grad():
        pe = net.point_eval(x.double(), d.double(), t.double()[:, None], with_color=color)
    # tolerances: a few x the reference's own fp32-vs-fp64 error on these quantities (tests/test_oracle_golden.py);
    # J / g_o / rgb go through the piecewise-constant ReLU Jacobian: quantile + loose max
    assert qd(ctx.view("xc"), pe["x_c"]) < 3e-6
    assert qd(ctx.view("sdf"), pe["sdf"]) < 1e-5
    assert qd(ctx.view("gc"), pe["g_c"]) < 1e-4
    if use_deform:
        jd = torch.einsum("mik,mk->mi", pe["J"], d.double())             # the kernels carry J d (JVP) and J^T g_c (VJP), not J
        assert qd(ctx.view("v"), jd, 0.99) < 5e-5
        assert qd(ctx.view("v"), jd) < 0.5
    assert qd(ctx.view("go"), pe["g_o"], 0.98) < 2e-4
    if color:
        assert qd(ctx.view("feat"), pe["feat"]) < 5e-5
        assert qd(ctx.view("rgb"), pe["rgb"], 0.98) < 5e-5
        assert qd(ctx.view("rgb"), pe["rgb"]) < 5e-2


@pytest.mark.parametrize("name", CASES)
def test_point_forward_golden(name):
    """Same kernels against per-point vectors captured from the reference itself."""
    from endosurf_amd import _lib
    c = load_case(name)
    use_deform = bool(c["meta/use_deform"])
    eng, flat, weff, packed, net = _setup(int(c["meta/seed"]), str(c["meta/mode"]), use_deform)
    x, d, t = (torch.from_numpy(c[k]).cuda().contiguous() for k in ("pt/x", "pt/d", "pt/t"))
    flags = (_lib.PF_DEFORM if use_deform else 0) | _lib.PF_COLOR
    ctx = eng.point_forward(eng.points(x=x, t=t.reshape(-1).contiguous(), dirs=d), weff, packed, flags)
    torch.cuda.synchronize()
    g = lambda k: torch.from_numpy(c[k])
    M = x.shape[0]
    assert qd(ctx.view("sdf"), g("pt64/sdf")) < 1e-5
    assert qd(ctx.view("feat"), g("pt64/feat")) < 5e-5
    assert qd(ctx.view("gc"), g("pt64/g_c")) < 1e-4
    assert qd(ctx.view("go"), g("pt64/g_o"), 0.98) < 2e-4
    assert qd(ctx.view("rgb"), g("pt64/rgb"), 0.98) < 5e-5
    if use_deform:
        assert qd(ctx.view("xc") - x, g("pt64/deform")) < 3e-6
        jd = torch.einsum("mik,mk->mi", g("pt64/J").reshape(M, 3, 3).double(), d.cpu().double())
        assert qd(ctx.view("v"), jd, 0.99) < 5e-5
