"""GPU parity of the hand-written backward: parameter gradients of the fused point evaluation and of the full
renderer / training loss against autograd on the fp64 oracle."""
import json
import os

import numpy as np
import pytest
import torch

import weightgen
from gpu_util import renderer_for, renderer_for_case
from oracle import endosurf_oracle as O
from oracle_util import CASES, RENDER_CFG, T, load_case, oracle_for

pytestmark = pytest.mark.gpu
LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _oracle(seed, mode, use_deform, dtype):
    state = weightgen.make_state(seed, mode, use_deform)
    params = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in state.items()}
    return O.OracleNet(params, use_deform), params


def _grad_table(r, params):
    """relative L2 error per parameter tensor: |g_hip - g_ref| / (|g_ref| + tiny)"""
    rows = {}
    named = dict(r.named_parameters())
    for k, p in params.items():
        ref = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().double().reshape(-1)
        got = named["model." + k].grad
        got = (got if got is not None else torch.zeros_like(named["model." + k])).detach().double().cpu().reshape(-1)
        rows[k] = (float((got - ref).norm() / (ref.norm() + 1e-30)), float(ref.norm()), float(got.norm()))
    return rows


def _dump(name, rows):
    os.makedirs(LOG, exist_ok=True)
    with open(os.path.join(LOG, f"grad_{name}.json"), "w") as f:
        json.dump(rows, f, indent=1)


@pytest.mark.parametrize("mode,use_deform,color", [("init", True, True), ("trained", True, True), ("trained", False, True),
                                                   ("trained", True, False)])
def test_point_backward(mode, use_deform, color):
    M = 200
    seed = 41
    r = renderer_for(seed, mode, use_deform)
    net, params = _oracle(seed, mode, use_deform, torch.float64)
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.uniform(-0.7, 0.7, size=(M, 3)).astype(np.float32))
    d = rng.normal(size=(M, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = torch.from_numpy(d.astype(np.float32))
    t = torch.from_numpy(rng.uniform(size=(M,)).astype(np.float32))
    ws, wg, wc = (torch.from_numpy(rng.normal(size=s).astype(np.float32)) for s in ((M, 1), (M, 3), (M, 3)))

    from endosurf_amd import _lib
    from endosurf_amd.renderer import _PointEvalFn
    weff, packed = r._weights()
    flags = r._flags(weff) | (_lib.PF_COLOR if color else 0)
    pts = r.engine.points(x=x.cuda(), t=t.cuda(), dirs=d.cuda())
    outs = _PointEvalFn.apply(weff, packed, r.engine, pts, flags)
    loss = (outs[0] * ws.cuda()).sum() + (outs[1] * wg.cuda()).sum()
    if color:
        loss = loss + (outs[2] * wc.cuda()).sum()
    loss.backward()
    torch.cuda.synchronize()

    pe = net.point_eval(x.double(), d.double(), t.double()[:, None], with_color=color)
    ref = (pe["sdf"] * ws.double()).sum() + (pe["g_o"] * wg.double()).sum()
    if color:
        ref = ref + (pe["rgb"] * wc.double()).sum()
    ref.backward()
    assert abs(float(loss) - float(ref)) < 2e-3 * max(1.0, abs(float(ref)))
    rows = _grad_table(r, params)
    _dump(f"point_{mode}_{int(use_deform)}_{int(color)}", rows)
    bad = {k: v for k, v in rows.items() if v[0] > 2e-2 and v[1] > 1e-7 and k != "deviation_network.variance"}
    assert not bad, bad
    med = float(np.median([v[0] for k, v in rows.items() if v[1] > 1e-7]))
    assert med < 2e-3, med
