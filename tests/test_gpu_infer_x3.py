"""GPU parity of the OPT-IN split-precision inference kernels (csrc/infer_x3r.hip: deformation value + tangent pass and VJP sweep, SDF value +
geometry features + reverse sweep on the bf16 matrix pipes with exactly split fp32 operands) -- held to the SAME budgets as the fp32 kernels
(tests/test_gpu_point.py) against the fp64 oracle and the per-point vectors captured from the reference."""
import numpy as np
import pytest
import torch

from oracle_util import CASES, load_case
from test_gpu_point import _setup, qd

pytestmark = pytest.mark.gpu


def _eval(eng, x, t, d, weff, packed, flags, split):
    eng.split_precision, eng.x3_infer_min = split, 1
    ctx = eng.point_forward(eng.points(x=x.cuda(), t=t.cuda(), dirs=d.cuda()), weff, packed, flags)
    torch.cuda.synchronize()
    return ctx


@pytest.mark.parametrize("mode,use_deform", [("init", True), ("trained", True), ("trained", False)])
@pytest.mark.parametrize("M,color", [(1, True), (100, True), (64, False), (777, True), (20000, True)])
def test_point_forward_x3(mode, use_deform, M, color):
    from endosurf_amd import _lib
    eng, flat, weff, packed, net = _setup(31, mode, use_deform)
    rng = np.random.default_rng(M)
    x = torch.from_numpy(rng.uniform(-0.8, 0.8, size=(M, 3)).astype(np.float32))
    d = rng.normal(size=(M, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = torch.from_numpy(d.astype(np.float32))
    t = torch.from_numpy(rng.uniform(size=(M,)).astype(np.float32))
    flags = (_lib.PF_DEFORM if use_deform else 0) | (_lib.PF_COLOR if color else 0)
    ctx = _eval(eng, x, t, d, weff, packed, flags, True)
    with torch.no_grad():
        pe = net.point_eval(x.double(), d.double(), t.double()[:, None], with_color=color)
    assert qd(ctx.view("xc"), pe["x_c"]) < 3e-6
    assert qd(ctx.view("sdf"), pe["sdf"]) < 1e-5
    assert qd(ctx.view("gc"), pe["g_c"]) < 1e-4
    if use_deform:
        jd = torch.einsum("mik,mk->mi", pe["J"], d.double())
        assert qd(ctx.view("v"), jd, 0.99) < 5e-5
        assert qd(ctx.view("v"), jd) < 0.5
    assert qd(ctx.view("go"), pe["g_o"], 0.98) < 2e-4
    if color:
        assert qd(ctx.view("feat"), pe["feat"]) < 5e-5
        assert qd(ctx.view("rgb"), pe["rgb"], 0.98) < 5e-5
        assert qd(ctx.view("rgb"), pe["rgb"]) < 5e-2
    # and against the fp32 kernels: the same points flip ReLU masks at most at isolated elements
    ref = _eval(eng, x, t, d, weff, packed, flags, False)
    assert qd(ctx.view("xc"), ref.view("xc")) < 3e-6
    assert qd(ctx.view("sdf"), ref.view("sdf")) < 1e-5
    assert qd(ctx.view("gc"), ref.view("gc")) < 1e-4
    if use_deform:
        assert qd(ctx.view("v"), ref.view("v"), 0.99) < 5e-5
    assert qd(ctx.view("go"), ref.view("go"), 0.98) < 2e-4


@pytest.mark.parametrize("name", CASES)
def test_point_forward_x3_golden(name):
    """The same kernels against per-point vectors captured from the reference itself."""
    from endosurf_amd import _lib
    c = load_case(name)
    use_deform = bool(c["meta/use_deform"])
    eng, flat, weff, packed, net = _setup(int(c["meta/seed"]), str(c["meta/mode"]), use_deform)
    x, d, t = (torch.from_numpy(c[k]).contiguous() for k in ("pt/x", "pt/d", "pt/t"))
    ctx = _eval(eng, x, t.reshape(-1).contiguous(), d, weff, packed, (_lib.PF_DEFORM if use_deform else 0) | _lib.PF_COLOR, True)
    g = lambda k: torch.from_numpy(c[k])
    M = x.shape[0]
    assert qd(ctx.view("sdf"), g("pt64/sdf")) < 1e-5
    assert qd(ctx.view("feat"), g("pt64/feat")) < 5e-5
    assert qd(ctx.view("gc"), g("pt64/g_c")) < 1e-4
    assert qd(ctx.view("go"), g("pt64/g_o"), 0.98) < 2e-4
    assert qd(ctx.view("rgb"), g("pt64/rgb"), 0.98) < 5e-5
    if use_deform:
        assert qd(ctx.view("xc") - x.cuda(), g("pt64/deform")) < 3e-6
        jd = torch.einsum("mik,mk->mi", g("pt64/J").reshape(M, 3, 3).double(), d.double())
        assert qd(ctx.view("v"), jd, 0.99) < 5e-5


def test_render_no_grad_split_precision_matches_fp32():
    """A no-grad render of 1024 rays with engine.split_precision: colour / depth maps agree with the fp32 render."""
    from gpu_util import renderer_for
    from endosurf_amd.trainer import SyntheticScene
    rays = SyntheticScene("cuda", seed=3).batch(1024)["rays"]
    outs = []
    for split in (False, True):
        r = renderer_for(24, "trained", True)
        r.engine.split_precision = split
        with torch.no_grad():
            outs.append(r(rays, iter_step=1, perturb_overwrite=False))
    a, b = outs
    dc = (a["color_map"] - b["color_map"]).abs()
    assert float(torch.quantile(dc.flatten(), 0.99)) < 2e-4 and float(dc.max()) < 5e-2
    dd = (a["depth_map"] - b["depth_map"]).abs()
    assert float(torch.quantile(dd.flatten(), 0.99)) < 2e-4 and float(dd.max()) < 5e-2


@pytest.mark.parametrize("use_deform", [True, False])
def test_split_inference_is_reproducible_at_full_size(use_deform):
    """BASELINE-sized batch (65 536 points = 1024 rays x 64 samples): three evaluations are bit-identical (a staging race in the
    weight / side streams would show up as isolated differing elements) and stay within rounding of the fp32 kernels."""
    from endosurf_amd import _lib
    M = 65536
    eng, flat, weff, packed, net = _setup(32, "trained", use_deform)
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.uniform(-0.8, 0.8, size=(M, 3)).astype(np.float32))
    d = rng.normal(size=(M, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = torch.from_numpy(d.astype(np.float32))
    t = torch.from_numpy(rng.uniform(size=(M,)).astype(np.float32))
    flags = (_lib.PF_DEFORM if use_deform else 0) | _lib.PF_COLOR
    keys = ("xc", "sdf", "gc", "go", "rgb", "feat")
    ref = _eval(eng, x, t, d, weff, packed, flags, False)
    R = {k: ref.view(k).clone() for k in keys}
    runs = []
    for _ in range(3):
        ctx = _eval(eng, x, t, d, weff, packed, flags, True)
        runs.append({k: ctx.view(k).clone() for k in keys})
    for k in keys:
        assert torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[0][k], runs[2][k]), k
    assert qd(runs[0]["xc"], R["xc"]) < 1e-6 and qd(runs[0]["sdf"], R["sdf"]) < 5e-6
    assert qd(runs[0]["gc"], R["gc"]) < 2e-5 and qd(runs[0]["feat"], R["feat"]) < 1e-5
    assert qd(runs[0]["go"], R["go"], 0.999) < 1e-3 and qd(runs[0]["rgb"], R["rgb"], 0.999) < 1e-4
