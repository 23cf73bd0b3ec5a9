"""CPU: host-side logic of the drop-in (no kernels run): schedules, config validation, the no-fallback rule, DP helpers."""
import math

import numpy as np
import pytest
import torch

from gpu_util import net_cfg
from oracle import endosurf_oracle as O
from oracle_util import RENDER_CFG


def test_lr_schedule_matches_oracle_restatement():
    from endosurf_amd.trainer import lr_factor
    for it in [0, 1, 100, 4999, 5000, 5001, 30000, 99999, 100000]:
        assert lr_factor(it) == pytest.approx(O.lr_factor(it), rel=0, abs=1e-15)
    assert lr_factor(2500) == 0.5 and lr_factor(100000) == pytest.approx(0.05)


def test_psnr_matches_oracle_restatement():
    from endosurf_amd.trainer import cal_psnr
    rng = np.random.default_rng(0)
    a, b = rng.uniform(size=(50, 3)).astype(np.float32), rng.uniform(size=(50, 3)).astype(np.float32)
    m = (rng.uniform(size=(50, 1)) > 0.3).astype(np.float32)
    got = float(cal_psnr(torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(m)))
    assert got == pytest.approx(O.cal_psnr(a, b, m), rel=1e-5)


def test_renderer_refuses_cpu_device_loudly():
    """The product path has no CPU / PyTorch fallback: constructing it off-GPU must raise, not degrade."""
    from endosurf_amd import EndoSurfRenderer
    from endosurf_amd._lib import EndoSurfHipError
    with pytest.raises(EndoSurfHipError):
        EndoSurfRenderer(dict(RENDER_CFG), net_cfg(True), device="cpu")


def test_product_never_imports_oracle():
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "endosurf_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_unsupported_architecture_is_rejected():
    from endosurf_amd.renderer import _check_arch
    cfg = net_cfg(True)
    _check_arch(cfg)
    cfg["sdf_network"]["hidden_dim"] = 128
    with pytest.raises(NotImplementedError):
        _check_arch(cfg)
    cfg = net_cfg(False)
    cfg["deform_network"]["hidden_dim"] = 3          # ignored when use_deform is False
    _check_arch(cfg)


def test_reference_style_init_statistics():
    from endosurf_amd import params
    from endosurf_amd.renderer import _reference_style_init
    from endosurf_amd import _lib
    lay = params.layout()
    flat = torch.zeros(int(_lib.load().es_param_floats()))
    torch.manual_seed(0)
    _reference_style_init(flat, lay, net_cfg(True))
    g = lambda k: flat[lay[k][0]:lay[k][0] + int(np.prod(lay[k][1]))].view(lay[k][1])
    W8, b8 = g("sdf_network.net.8.weight_v"), g("sdf_network.net.8.bias")
    assert abs(float(W8.mean()) - math.sqrt(math.pi) / 16) < 1e-4 and float(b8[0]) == pytest.approx(-0.8)
    W0 = g("sdf_network.net.0.weight_v")
    assert float(W0[:, 3:].abs().max()) == 0.0 and float(W0[:, :3].std()) == pytest.approx(math.sqrt(2) / 16, rel=0.1)
    W4 = g("sdf_network.net.4.weight_v")
    assert float(W4[:, -36:].abs().max()) == 0.0
    for net in ("deform_network", "sdf_network", "color_network"):
        for l in range(9):
            v, gg = g(f"{net}.net.{l}.weight_v"), g(f"{net}.net.{l}.weight_g")
            assert torch.allclose(gg[:, 0], v.norm(dim=1), rtol=1e-6)       # weight_norm: g = ||v|| at init
    assert float(g("deviation_network.variance")) == pytest.approx(0.3)
    # geometric init makes sdf(x) ~ |x| - 0.8: check through the oracle
    state = {k: g(k).clone() for k in lay}
    net = O.OracleNet(state, True)
    x = torch.tensor([[0.3, 0.2, -0.1], [0.5, 0.0, 0.0], [0.0, -0.9, 0.0]])   # away from 0: softplus(0) = ln2/100 offsets add up there
    with torch.no_grad():
        sdf, _, _ = net.sdf_net(x, with_grad=False)
    assert torch.allclose(sdf[:, 0], x.norm(dim=1) - 0.8, atol=0.15)


def test_synthetic_scene_rays_hit_the_unit_sphere():
    from endosurf_amd.trainer import SyntheticScene
    b = SyntheticScene("cpu", seed=3).batch(256)
    rays = b["rays"]
    assert rays.shape == (256, 9) and torch.allclose(rays[:, 3:6].norm(dim=1), torch.ones(256), atol=1e-6)
    near, far = O.sphere_intersection(rays[:, :3], rays[:, 3:6])
    assert bool((far > near).all()) and bool((rays[:, 8] == rays[0, 8]).all())
    assert b["color"].shape == (256, 3) and float(b["depth"].min()) >= 1.2
