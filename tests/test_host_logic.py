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


# ---- round 2 host logic -------------------------------------------------------------------------------------------------
def _cpu_model(use_deform=True):
    """The parameter container alone (flat buffer + reference-named views) lives happily on the CPU; only kernels need the GPU."""
    from types import SimpleNamespace
    from endosurf_amd.renderer import EndoSurfNet
    torch.manual_seed(3)
    m = EndoSurfNet(net_cfg(use_deform), "cpu")
    m._pack_cache, m._flat_grad, m._epoch = None, None, 0
    return m, SimpleNamespace(model=m, engine=None)


@pytest.mark.parametrize("use_deform", [True, False])
def test_flat_adam_state_dict_round_trips_through_torch_adam(use_deform):
    """FlatAdam.state_dict() IS a torch.optim.Adam state_dict (the reference's ckpt["optimizer"], trainer_endosurf.py:76-92) and
    FlatAdam.load_state_dict() accepts one: the flat moment buffers are scattered / gathered in get_train_params() order."""
    from endosurf_amd.trainer import FlatAdam
    m, r = _cpu_model(use_deform)
    groups = m.get_train_params()
    plist = [p for k in groups for p in groups[k]]
    assert len(plist) == (82 if use_deform else 55)          # 27 tensors per network + the variance
    opt = torch.optim.Adam(plist, lr=5e-4)
    g = torch.Generator().manual_seed(0)
    for _ in range(2):
        for p in plist:
            p.grad = torch.randn(p.shape, generator=g)
        opt.step()
    sd = opt.state_dict()
    fa = FlatAdam(r)
    fa.load_state_dict(sd)
    assert fa.step_count == 2 and fa.param_groups[0]["lr"] == 5e-4
    lay = m._layout
    off, shape = lay["sdf_network.net.4.weight_v"]
    idx = [i for i, p in enumerate(plist) if p is m.sdf_network.net[4].weight_v][0]
    assert torch.equal(fa.exp_avg[off:off + 256 * 295].view(256, 295), sd["state"][idx]["exp_avg"])
    voff = lay["deviation_network.variance"][0]
    assert torch.equal(fa.exp_avg_sq[voff], sd["state"][len(plist) - 1]["exp_avg_sq"])
    if not use_deform:                                       # the deformation slots of the flat buffer stay untouched
        d0 = lay["deform_network.net.0.bias"][0]
        assert float(fa.exp_avg[d0:lay["sdf_network.net.0.bias"][0]].abs().max()) == 0.0
    # back: a fresh torch Adam accepts FlatAdam's state_dict and holds identical moments
    sd2 = fa.state_dict()
    assert set(sd2["param_groups"][0]) == set(sd["param_groups"][0])
    opt2 = torch.optim.Adam(plist, lr=1.0)
    opt2.load_state_dict(sd2)
    for p in plist:
        a, b = opt.state[p], opt2.state[p]
        assert torch.equal(a["exp_avg"], b["exp_avg"]) and torch.equal(a["exp_avg_sq"], b["exp_avg_sq"]) and float(a["step"]) == float(b["step"])
    assert opt2.param_groups[0]["lr"] == 5e-4
    # the flat round-1 format still loads
    fa2 = FlatAdam(r)
    fa2.load_state_dict(dict(step=2, exp_avg=fa.exp_avg.clone(), exp_avg_sq=fa.exp_avg_sq.clone(), param_groups=[dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8)]))
    assert fa2.step_count == 2 and torch.equal(fa2.exp_avg, fa.exp_avg)
    with pytest.raises(ValueError):
        bad = {"state": {}, "param_groups": [dict(sd["param_groups"][0], params=list(range(5)))]}
        fa2.load_state_dict(bad)


def test_rebound_parameters_are_folded_back_into_the_flat_buffer():
    m, _ = _cpu_model(True)
    p = m.color_network.net[2].bias
    off = m._layout["color_network.net.2.bias"][0]
    m._check_views()                                           # all views intact: nothing happens
    e0 = m._epoch
    p.data = torch.full_like(p.data, 0.5)                      # a loader re-binds the storage
    assert p.data_ptr() != m._flat.data_ptr() + 4 * off
    m._check_views()
    assert p.data_ptr() == m._flat.data_ptr() + 4 * off and float(m._flat[off:off + 256].min()) == 0.5 and m._epoch > e0
    with pytest.raises(TypeError):
        m.double()                                             # fp32 on an AMD GPU only: never silently converted


def test_bench_work_model_and_self_launch(monkeypatch):
    import importlib
    import bench
    importlib.reload(bench)
    c2, c4 = bench.CONFIGS[2], bench.CONFIGS[4]
    a2, a4 = bench.algorithmic_gflop_per_ray(c2), bench.algorithmic_gflop_per_ray(c4)
    D, S, C = bench.MAC_D, bench.MAC_S, bench.MAC_C
    assert a2["upsample"] * 1e9 == pytest.approx(2 * 56 * (D + S)) and a2["render_core_forward"] * 1e9 == pytest.approx(2 * 64 * (3 * D + 2 * S + C))
    assert a4["render_core_forward"] * 1e9 == pytest.approx(2 * 64 * (2 * S + C))          # no deformation network
    assert bench.kernel_macs("k_query_sdf", False, executed=False) == S and bench.kernel_macs("k_wgrad[deform]", True) == 3 * D
    # executed work: a query issues only the sdf row of the [257 x 256] last layer; the SDF chain kernels skip it in one of their sweeps
    assert bench.kernel_macs("k_query_sdf", True) == D + S - 65536 and bench.kernel_macs("k_query_sdf_x3", True) == D + S - 65536
    assert bench.kernel_macs("k_sdf_fwd", True) == 2 * S - 65536 and bench.kernel_macs("k_sdf_bwd", True) == 2 * S - 65792
    assert bench.kernel_macs("k_wgrad_x3[sdf]", True) == 2 * S - 65536 and bench.kernel_macs("k_color_fwd", True, executed=False) == C
    assert bench.kernel_macs("k_query_sdf[later marching blocks: tiles of finished rays exit]", True) is None      # never counted as work
    assert bench.render_cfg(bench.CONFIGS[3])["n_samples"] == 64 and bench.CONFIGS[3]["rays"] == 2048
    # `python bench.py --gpus 4` without a launcher in the environment re-executes itself under torch.distributed.run
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("ES_DIST_BACKEND", raising=False)
    # fewer GPUs than ranks: exits at once with the numbers in the message, nothing is launched (it used to fold ranks onto one GPU)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "only 2 GPU(s) visible" in str(e.value.code) and "cmd" not in seen
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_workspace_budget_chunking_arithmetic():
    """_chunk_rays: rays per chunk of a grad-enabled render under render_cfg["workspace_gb"] (host arithmetic over the C layout query)."""
    from types import SimpleNamespace
    from endosurf_amd import _lib
    from endosurf_amd.renderer import EndoSurfRenderer
    lib = _lib.load()
    flags = _lib.PF_DEFORM | _lib.PF_SAVE
    per_point = 4.0 * lib.es_point_workspace_floats(65536, flags | _lib.PF_COLOR) / 65536
    assert 90e3 < per_point < 120e3                                  # ~103 KB of saved activations per point
    fake = SimpleNamespace(workspace_gb=64.0, engine=SimpleNamespace(lib=lib), __dict__={})
    f = lambda N, S, fl=flags: EndoSurfRenderer._chunk_rays(fake, N, S, fl)
    assert f(1024, 64) == 0 and f(2048, 128) == 0                     # configs 2 and 3 fit
    c = f(327680, 64)                                                 # a full frame under grad: chunked
    assert c > 0 and c % 64 == 0 and c * 64 * per_point <= 64e9 < (c + 64) * 64 * per_point
    assert f(327680, 64, _lib.PF_DEFORM) == 0                         # no saved activations: no budget applies
    fake.workspace_gb = 1e-3
    fake.__dict__.pop("_budget_points", None)
    with pytest.raises(_lib.EndoSurfHipError, match="workspace_gb"):
        f(1024, 64)


def test_philox_reference_known_answers():
    """Random123's known-answer vectors for philox4x32-10 (kat_vectors): pins tests/philox_ref.py, the checker of es_uniform."""
    from philox_ref import philox4x32_10
    assert philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_rank_pinning_gives_disjoint_core_sets_within_the_allowed_mask():
    """parallel.pin_rank_to_cores (bench.py --gpus N, cold-run kit): every local rank gets its own contiguous run of the cores this
    process may use -- never a core outside that mask -- and fewer cores than ranks means nothing is pinned (and the result says why)."""
    import os
    from endosurf_amd import parallel
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_getaffinity")
    allowed = sorted(os.sched_getaffinity(0))
    threads = torch.get_num_threads()
    try:
        world = 2 if len(allowed) >= 2 else 1
        got = []
        for local in range(world):
            os.sched_setaffinity(0, allowed)
            info = parallel.pin_rank_to_cores(local, world)
            mine = sorted(os.sched_getaffinity(0))
            got.append((info, mine))
        if world == 1:
            assert got[0][0]["pinned"] is False and "reason" in got[0][0]
        else:
            (i0, c0), (i1, c1) = got
            assert i0["pinned"] and i1["pinned"] and not (set(c0) & set(c1)) and set(c0) | set(c1) <= set(allowed)
            assert i0["cores"] == len(c0) == len(allowed) // 2 and (i0["first"], i0["last"]) == (c0[0], c0[-1]) and c0[-1] < c1[0]
        os.sched_setaffinity(0, allowed)
        too_many = parallel.pin_rank_to_cores(0, len(allowed) + 1)
        assert too_many["pinned"] is False and str(len(allowed)) in too_many["reason"] and sorted(os.sched_getaffinity(0)) == allowed
    finally:
        os.sched_setaffinity(0, allowed)
        torch.set_num_threads(threads)


def test_pmc_figures_are_per_single_launch_and_plausible():
    """bench.pmc_info / pmc_rates: a symbol with several (instantiation, grid) rows -- k_query_sdf's marching and
    coarse launches -- is reported per SINGLE launch (launch-weighted mean), and rates that no MI355X can show are withheld, not printed."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    pm = bench.pmc_info("k_query_sdf")
    assert pm is not None and len(pm["rows"]) >= 2
    per = [r["hbm_bytes"] for r in pm["rows"]]
    assert min(per) <= pm["hbm_bytes"] <= max(per) and 30e6 < pm["hbm_bytes"] < 40e6          # ~34 MB: 8 L2 fills of the weight pack
    n = sum(r["launches"] for r in pm["rows"])
    assert abs(pm["cycles"] - sum(r["cycles"] * r["launches"] for r in pm["rows"]) / n) < 1
    gbps, ghz, note = bench.pmc_rates(pm, 1.283)
    assert note is None and 1.0 <= ghz <= 2.6 and gbps < 100
    assert bench.pmc_rates(pm, 0.6)[2] is not None and bench.pmc_rates(pm, 0.6)[:2] == (None, None)       # 4.7 GHz: withheld


def test_device_table_slots_are_independent(tmp_path):
    """csrc/device_table.h (the per-device state of the opt-in kernel timers, csrc/timing.hip): every device ordinal has its own slot,
    out-of-range ordinals stay in bounds.  Plain C++: compiled with g++ and run here."""
    import os
    import shutil
    import subprocess
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    src = tmp_path / "t.cpp"
    src.write_text('''
#include <vector>
#include <cstdio>
#include "device_table.h"
struct S { bool on = false; std::vector<int> rec; };
int main() {
    static es::DeviceTable<S> t;
    t.at(0).on = true; t.at(0).rec.push_back(7);
    t.at(3).rec.push_back(9); t.at(3).rec.push_back(10);
    if (t.at(1).on || !t.at(1).rec.empty()) return 1;           // untouched device: default state
    if (!t.at(0).on || t.at(0).rec.size() != 1 || t.at(3).on || t.at(3).rec.size() != 2) return 2;
    t.at(0).rec.clear();                                          // draining device 0 leaves device 3 alone
    if (t.at(3).rec.size() != 2) return 3;
    if (&t.at(-5) != &t.at(0) || &t.at(1000) != &t.at(63)) return 4;   // out-of-range ordinals: clamped, never out of bounds
    std::puts("ok");
    return 0;
}
''')
    exe = tmp_path / "t"
    inc = os.path.join(REPO, "endosurf_amd", "csrc")
    subprocess.run([gxx, "-std=c++17", "-I", inc, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.returncode, out.stdout, out.stderr)


def test_hbm_accounting_against_the_committed_counters():
    """bench.py's algorithmic HBM bytes per launch (HBM_BYTES_PER_POINT x the points of a launch: what the stream-heavy kernels MUST move)
    beside the committed rocprofv3 traffic of the same symbols (profiles/*_pmc_summary.json): the chain kernels move at most 10 % more than
    they must, the weight-gradient GEMMs carry the documented second read of dA (DEAD_ENDS C5) and say so."""
    import os
    import sys
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, REPO)
    import bench
    rows = {r["kernel"]: r for r in bench.hbm_accounting(65536, 3072)}
    assert set(rows) == {"k_deform_fwd", "k_sdf_fwd", "k_color_fwd", "k_deform_vjp", "k_color_bwd", "k_deform_tan", "k_sdf_bwd", "k_deform_bwd",
                         "k_wgrad[deform]", "k_wgrad[sdf]", "k_wgrad[color]"}
    for k, r in rows.items():
        assert r["pmc_hbm_bytes_per_launch"] and r["algorithmic_bytes_per_launch"] > 0, k
        if k.startswith("k_wgrad"):
            assert 1.0 <= r["ratio"] <= 1.6 and "dA" in r["explanation"], (k, r["ratio"])
        else:
            assert 0.98 <= r["ratio"] <= 1.10 and r["explanation"] is None, (k, r["ratio"])
    # the two-sweep floor of the SDF backward: 7 streams x 8 layers x 1 KiB per point
    assert bench.HBM_BYTES_PER_POINT["sdf_bwd"] >= 7 * 8 * 1024
