"""The OPT-IN split-precision mode (ES_SPLIT_BF16=1 / engine.split_precision: large SDF queries and the weight-gradient GEMMs on the
bf16 matrix pipes with every fp32 operand split exactly into three bf16 planes) must pass the SAME parity checks, with the SAME
budgets, as the fp32 product path: the parameter-gradient tests of test_gpu_backward.py are re-run under the switch."""
import pytest
import torch

import test_gpu_backward as B
from oracle_util import CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _split(monkeypatch):
    monkeypatch.setenv("ES_SPLIT_BF16", "1")


def _assert_on():
    from gpu_util import renderer_for
    assert renderer_for(1, "init", True).engine.split_precision


@pytest.mark.parametrize("mode,use_deform,color", [("init", True, True), ("trained", True, True), ("trained", False, True), ("trained", True, False)])
def test_point_backward_split(mode, use_deform, color):
    _assert_on()
    B.test_point_backward(mode, use_deform, color)


@pytest.mark.parametrize("name", CASES)
def test_render_scalar_param_grads_split(name):
    B.test_render_scalar_param_grads(name)


@pytest.mark.parametrize("name", CASES)
def test_training_loss_param_grads_split(name):
    B.test_training_loss_param_grads(name)


def test_split_wgrad_close_to_fp32_wgrad():
    """Same activations, same adjoints: the split-precision weight gradient equals the fp32-MFMA one to fp32 rounding (1e-5 relative
    per tensor; the atomics' summation order alone moves the fp32 path by ~1e-6), in atomic and in deterministic mode."""
    from gpu_util import renderer_for
    from endosurf_amd.trainer import SyntheticScene, compute_loss_fused
    b = SyntheticScene("cuda", seed=8).batch(1024)
    u, un = torch.rand(1024, 1, device="cuda"), torch.rand(1024, 3, device="cuda")
    grads = {}
    for split, det in ((False, True), (True, True), (True, False)):
        r = renderer_for(24, "trained", True)
        r.engine.split_precision = False           # queries in fp32 for all runs: identical sample positions
        r.engine.deterministic = det
        total, _, _ = compute_loss_fused(r, b, 1, u_perturb=u, u_neigh=un)
        r.engine.split_precision = split           # only the weight-gradient GEMMs differ
        total.backward()
        torch.cuda.synchronize()
        grads[(split, det)] = {k: p.grad.double().clone() for k, p in r.named_parameters()}
    ref = grads[(False, True)]
    for key in ((True, True), (True, False)):
        for k, g in grads[key].items():
            n = float(ref[k].norm())
            assert float((g - ref[k]).norm()) <= 1e-5 * n + 1e-9, (key, k, float((g - ref[k]).norm()), n)
