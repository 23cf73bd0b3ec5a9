"""bench.py's N > 1 path end to end on ONE GPU: two ranks (torch.distributed.run, gloo instead of RCCL because both ranks
share the device) run the data-parallel training bench — broadcast, flat-gradient all-reduce, Adam with the 1/world scale —
and rank 0 prints the JSON line.  Guards against deadlocks in the post-timing sections (only some ranks stepping)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode", ["train", "frame"])
def test_bench_two_ranks_one_gpu(mode):
    env = dict(os.environ, ES_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2" if mode == "train" else "1",
           "--warmup", "1" if mode == "train" else "0", "--no-cpu-baseline", "--mode", mode]          # (a frame step = 327 680 rays: one is enough)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]              # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == (2 if mode == "train" else 1) and d["value"] > 0
    # what the collective itself saw: an all-reduce of ones, every rank's own time, the bucket's all-reduce time
    assert d["ranks_seen_by_collective"] == 2 and 0 < d["ms_per_step_min"] <= d["ms_per_step_max"] and d["allreduce_ms"] > 0
    if mode == "train":
        assert d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2" and d["roofline"]["frac"] > 0
        # the default workload at N > 1 appends the multi-GPU configurations of BASELINE.json as extras
        ex = d["extras"]
        assert set(ex) >= {"cfg4", "cfg5_frame"} and "forward" not in ex, ex.keys()
        for k in ("cfg4", "cfg5_frame"):
            assert "error" not in ex[k], ex[k]
            assert ex[k]["value"] > 0 and ex[k]["n_gpus"] == 2 and 0 < ex[k]["roofline"]["frac"] < 1, ex[k]
    else:
        assert d["extras"] is None and d["config"]["frame_rows_per_rank"] == [256, 256] and d["config"]["rays_per_step_whole_job"] == 512 * 640


def test_bench_eight_ranks_one_gpu_cold_run_kit():
    """The line of an 8-rank run must be diagnosable: eight ranks (gloo, one GPU, a 64-ray batch each) -- the
    collective saw all eight, every rank reports its host issue time per step, its GPU-idle estimate (eager step - the same step replayed
    from the whole-step hipGraph) and the host cores it was pinned to, the fallback rule was evaluated, and after all steps the eight
    replicas hold bit-identical parameters."""
    env = dict(os.environ, ES_DIST_BACKEND="gloo", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--rays", "64", "--headline-only"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "dp8" and d["config"]["rays_per_step_whole_job"] == 8 * 64 and d["value"] > 0
    assert d["ranks_seen_by_collective"] == 8
    assert d["collective_proof"]["replicas_bit_identical"] is True, d["collective_proof"]
    pr = d["per_rank"]
    assert sorted(r["rank"] for r in pr) == list(range(8))
    for r in pr:
        assert r["host_issue_ms"] > 0 and r["gpu_idle_ms"] >= 0 and r["ms_per_step"] > 0 and "pinned" in r["affinity"], r
    pinned = [r["affinity"] for r in pr if r["affinity"]["pinned"]]
    if pinned:          # (fewer than 8 usable host cores: nothing is pinned, and the line says why)
        assert len(pinned) == 8
        spans = sorted((a["first"], a["last"]) for a in pinned)
        assert all(spans[i][1] < spans[i + 1][0] for i in range(7)), spans          # disjoint core sets
    else:
        assert all("reason" in r["affinity"] for r in pr)
    assert 0 < d["host_issue_ms_min"] <= d["host_issue_ms_max"] and 0 <= d["gpu_idle_ms_min"] <= d["gpu_idle_ms_max"]
    gp = d["graph_probe"]
    assert "error" not in gp and gp["ms_per_step"] > 0 and gp["use_graph"] == (gp["host_bound"] and gp["ms_per_step"] < gp["eager_ms_per_step"]), gp
    assert d["config"]["whole_step_hipgraph"] == gp["use_graph"]
    assert d["allreduce_ms"] > 0 and d["collective_proof"]["allreduce_exposed_ms"] > 0
    om = d["collective_proof"]["allreduce_other_mode"]          # a few steps of the pipelined all-reduce next to the one-bucket headline
    assert om["overlap_allreduce"] is True and om["pipelined_steps"] >= 8 and om["ms_per_step"] > 0, om
    assert d["collective_proof"]["replicas_bit_identical"] is True


def test_bench_self_launches_without_torchrun():
    """``python bench.py --gpus 2`` with no RANK / WORLD_SIZE in the environment must start its own ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ES_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["parallelism"] == "dp2"
    assert d["roofline"]["end_to_end"]["frac"] > 0 and d["config"]["with_early_exit"]["value"] > 0
    assert d["ranks_seen_by_collective"] == 2


def test_bench_default_line_carries_extras():
    """The ONE command the driver runs (``python bench.py``; here with fewer steps and without the CPU baseline) prints the headline AND
    every other number the builder reports: forward-only, split-precision training, cfg3, cfg4, one cfg5 frame."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["baseline_config"] == 2 and d["dtype"] == "f32" and d["n_gpus"] == 1 and d["ranks_seen_by_collective"] is None
    ex = d["extras"]
    assert set(ex) == {"reference_call_sequence", "forward", "split_precision_train", "cfg3", "cfg4", "cfg5_frame"}, ex.keys()
    # the reference trainer's own loop through the drop-in: without and with its logging traffic; from its second step
    # on the two auxiliary calls' points live in the render's workspace (3 x 1024 rows)
    rs = ex.pop("reference_call_sequence")
    assert "error" not in rs, rs
    for k in ("plain", "with_reference_logging", "plain_with_flat_adam"):
        e = rs[k]
        assert e["ms_per_step"] > 0 and e["value"] > 0 and e["sum_timed_kernel_ms"] > 0 and e["aux_rows_in_render_workspace"] == 3072, (k, e)
        assert e["with_early_exit"]["ms_per_step"] > 0
    assert rs["plain"]["host_syncs_per_step"] == 1 and rs["with_reference_logging"]["host_syncs_per_step"] == 14
    assert rs["plain"]["host_issue_ms"] < rs["plain"]["ms_per_step"] < 1.5 * d["ms_per_step"]
    for k, e in ex.items():
        assert "error" not in e, (k, e)
        assert e["value"] > 0 and e["ms_per_step"] > 0 and 0 < e["roofline"]["frac"] < 1 and e["roofline"]["end_to_end"]["frac"] > 0, (k, e)
    # forward-only is faster than a training step; split precision is faster than fp32; cfg3 is the big one
    assert ex["forward"]["ms_per_step"] < d["ms_per_step"] and ex["split_precision_train"]["ms_per_step"] < d["ms_per_step"] < ex["cfg3"]["ms_per_step"]
    sp = ex["split_precision_train"]["roofline"]
    assert "_x3" in sp["kernel"] and sp["peak"] == 2500.0            # the mode's roofline is taken on a bf16-pipe kernel
    assert sp["pipes"]["bf16"]["ms"] > 0 and sp["pipes"]["fp32"]["ms"] > 0 and 0 < sp["pipes"]["bf16"]["frac_of_2500"] < 1
    assert "1 x 2500" not in json.dumps(sp)


@pytest.mark.parametrize("config", [3, 4])
def test_bench_other_baseline_configs(config):
    """--config 3 (2048 rays x 128 samples) and --config 4 (use_deform False) run and report a roofline."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--config", str(config), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["baseline_config"] == config and d["value"] > 0 and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["use_deform"] == (config != 4)
    assert d["config"]["samples_per_ray"] == (128 if config == 3 else 64)


def test_data_parallel_ranks_stay_identical():
    """Two ranks with different batches: after the flat all-reduce + FlatAdam the replicas hold identical parameters, and they
    equal a single process that averages the two gradients itself."""
    code = r'''
import os, sys, json
sys.path.insert(0, os.environ["ES_REPO"]); sys.path.insert(0, os.path.join(os.environ["ES_REPO"], "tests"))
import torch
from endosurf_amd import parallel
from endosurf_amd.trainer import Trainer, SyntheticScene
from gpu_util import renderer_for
rank, world, local = parallel.init_distributed("gloo")
torch.cuda.set_device(0)
r = renderer_for(5, "trained", True)
tr = Trainer(r, lr=1e-3, data_parallel=True)
parallel.broadcast_parameters(tr.params)
sc = SyntheticScene("cuda", seed=100 + rank)
for it in range(2):
    tr.update_learning_rate(7000 + it)
    tr.train_step(sc.batch(256), 20000 + it)
flat = r.model._flat.detach().clone()
both = [torch.zeros_like(flat) for _ in range(world)]
torch.distributed.all_gather(both, flat)
if rank == 0:
    print(json.dumps(dict(maxdiff=float((both[0] - both[1]).abs().max()), moved=float((flat - torch.load(os.environ["ES_INIT"]).cuda()).abs().max()))))
torch.distributed.destroy_process_group()
'''
    import tempfile
    import torch
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from gpu_util import renderer_for
    with tempfile.TemporaryDirectory() as td:
        init = os.path.join(td, "init.pt")
        torch.save(renderer_for(5, "trained", True).model._flat.detach().cpu(), init)
        script = os.path.join(td, "w.py")
        open(script, "w").write(code)
        env = dict(os.environ, ES_REPO=REPO, ES_INIT=init)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), script]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["maxdiff"] == 0.0, d                    # same summed bucket, same update on every rank
    assert d["moved"] > 1e-4, d


@pytest.mark.parametrize("use_deform", [True, False])
def test_pipelined_allreduce_equals_the_single_bucket(use_deform):
    """Trainer(overlap_allreduce=True): the gradient all-reduce as three buckets, two of them issued on a side stream behind the
    weight-gradient launch that completes them (es_point_backward_stages + es_weightnorm_backward_layers).  Two ranks with different
    batches (gloo, one GPU), deterministic reductions: after three steps the parameters equal those of the default single-bucket step
    bit for bit, and the replicas are identical."""
    code = r'''
import os, sys, json
sys.path.insert(0, os.environ["ES_REPO"]); sys.path.insert(0, os.path.join(os.environ["ES_REPO"], "tests"))
import torch
from endosurf_amd import parallel
from endosurf_amd.trainer import Trainer, SyntheticScene
from gpu_util import renderer_for
rank, world, local = parallel.init_distributed("gloo")
torch.cuda.set_device(0)
use_deform = os.environ["ES_USE_DEFORM"] == "1"
def run(overlap):
    torch.manual_seed(0)
    r = renderer_for(5, "trained", use_deform)
    r.engine.deterministic = True
    tr = Trainer(r, lr=1e-3, data_parallel=True, overlap_allreduce=overlap)
    parallel.broadcast_parameters(tr.params)
    sc = SyntheticScene("cuda", seed=100 + rank)
    gen = torch.Generator(device="cuda"); gen.manual_seed(50 + rank)
    for it in range(3):
        b = sc.batch(256)
        u, un = torch.rand(256, 1, device="cuda", generator=gen), torch.rand(256, 3, device="cuda", generator=gen)
        tr.update_learning_rate(7000 + it)
        tr.train_step(b, 20000 + it, u_perturb=u, u_neigh=un)
    torch.cuda.synchronize()
    assert tr.pipelined_steps == (3 if overlap else 0), tr.pipelined_steps
    return r.model._flat.detach().clone()
a = run(False)
b = run(True)
both = [torch.zeros_like(b) for _ in range(world)]
torch.distributed.all_gather(both, b)
if rank == 0:
    print(json.dumps(dict(overlap_vs_single=int((a != b).sum()), replicas=int((both[0] != both[1]).sum()),
                          moved=float((a - renderer_for(5, "trained", use_deform).model._flat).abs().max()))))
torch.distributed.destroy_process_group()
'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        script = os.path.join(td, "w.py")
        open(script, "w").write(code)
        env = dict(os.environ, ES_REPO=REPO, ES_USE_DEFORM="1" if use_deform else "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), script]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["overlap_vs_single"] == 0 and d["replicas"] == 0 and d["moved"] > 1e-4, d


def test_rccl_world1_trainer_allreduce():
    """The RCCL path itself, on one GPU: backend "nccl" (= RCCL on ROCm) at world size 1, parameter broadcast, and a data-parallel
    Trainer whose flat 6.6 MB gradient bucket goes THROUGH the all-reduce (``force_collective``): librccl loads, the communicator
    initialises, the collective runs on the bucket FlatAdam consumes.  Sum over one rank = identity, so the trained parameters must
    equal those of a non-distributed run bit for bit (deterministic reductions)."""
    code = r'''
import os, sys, json
sys.path.insert(0, os.environ["ES_REPO"]); sys.path.insert(0, os.path.join(os.environ["ES_REPO"], "tests"))
import torch
import torch.distributed as dist
from endosurf_amd import parallel
from endosurf_amd.trainer import Trainer, SyntheticScene
from gpu_util import renderer_for
rank, world, local = 0, 1, 0
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
def run(dp, overlap=False):
    torch.manual_seed(0)                     # the stratified jitter / neighbour offsets are drawn with torch.rand on the device
    r = renderer_for(5, "trained", True)
    r.engine.deterministic = True
    tr = Trainer(r, lr=1e-3, data_parallel=dp, force_collective=dp, overlap_allreduce=overlap)
    if dp:
        parallel.broadcast_parameters(tr.params)
    sc = SyntheticScene("cuda", seed=100)
    for it in range(2):
        tr.update_learning_rate(7000 + it)
        tr.train_step(sc.batch(256), 20000 + it)
    return r.model._flat.detach().clone()
a = run(True)
g = torch.full((1654951,), 2.0, device="cuda")
w = parallel.allreduce_flat(g, force=True)
torch.cuda.synchronize()
b = run(False)
c = run(True, overlap=True)                  # the bucket pipeline: its side-stream all-reduces go through RCCL as well
print(json.dumps(dict(world=w, bucket_ok=bool((g == 2.0).all()), maxdiff=float((a - b).abs().max()), finite=bool(torch.isfinite(a).all()),
                      maxdiff_pipelined=float((c - b).abs().max()))))
dist.destroy_process_group()
'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        script = os.path.join(td, "w.py")
        open(script, "w").write(code)
        env = dict(os.environ, ES_REPO=REPO, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["world"] == 1 and d["bucket_ok"] and d["finite"], d
    assert d["maxdiff"] == 0.0 and d["maxdiff_pipelined"] == 0.0, d


def test_bench_rccl_world1_line():
    """bench.py under torch.distributed.run with ONE rank and the default backend: the whole N-rank code path (RCCL rendezvous,
    broadcast, barrier, MAX all-reduce of the time) runs on the box's single GPU and prints its JSON line."""
    env = {k: v for k, v in os.environ.items() if k != "ES_DIST_BACKEND"}
    env["ES_FORCE_DIST"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["collective"] == "rccl all-reduce forced at world 1"
    assert d["ranks_seen_by_collective"] == 1 and d["allreduce_ms"] > 0 and d["collective_proof"]["bucket_bytes"] == 4 * 1654951
    assert not [k for k, e in d["extras"].items() if isinstance(e, dict) and "error" in e], d["extras"]      # the extras' collectives ran through RCCL too


def test_bench_more_ranks_than_gpus_fails_fast():
    """``--gpus N`` with fewer than N GPUs on the node exits at once with a clear message (it used to fold ranks onto one device)."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ES_DIST_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode != 0
    assert f"only {n - 1} GPU(s) visible" in out.stderr, out.stderr[-2000:]


def test_exact_denominators_two_ranks_equal_one_big_batch():
    """Trainer(exact_denominators=True): 2 ranks x 512 rays give, after the summing all-reduce and the 1/world scale, the gradient of ONE
    rank with the 1024-ray batch (the <= 6 loss normalisers all-reduced before the backward) -- with masks that differ between
    the halves, so that the standard per-rank ratios would NOT agree (also checked)."""
    code = r'''
import os, sys, json
sys.path.insert(0, os.environ["ES_REPO"]); sys.path.insert(0, os.path.join(os.environ["ES_REPO"], "tests"))
import torch
import torch.distributed as dist
from endosurf_amd import parallel
from endosurf_amd.trainer import Trainer, SyntheticScene
from gpu_util import renderer_for
rank, world, local = parallel.init_distributed("gloo")
torch.cuda.set_device(0)
N = 1024
sc = SyntheticScene("cuda", seed=42)
gen = torch.Generator(device="cuda"); gen.manual_seed(9)
b = sc.batch(N)
b["mask"] = (torch.rand(N, 1, device="cuda", generator=gen) < 0.7).float()
b["mask"][:N // 2] *= (torch.rand(N // 2, 1, device="cuda", generator=gen) < 0.4).float()        # the halves differ a lot
b["color_mask"] = (torch.rand(N, 1, device="cuda", generator=gen) < 0.8).float()
u, un = torch.rand(N, 1, device="cuda", generator=gen), torch.rand(N, 3, device="cuda", generator=gen)
def grad(sl, exact, dp):
    r = renderer_for(5, "trained", True)
    r.engine.deterministic = True
    tr = Trainer(r, data_parallel=dp, exact_denominators=exact)
    tr.optimizer.zero_grad()
    loss, _, _ = tr.loss_fn(r, {k: v[sl] for k, v in b.items()}, 2000, tr.loss_weights, tr.surf_neig_rad, u[sl], un[sl])
    loss.backward()
    g = tr.optimizer.flat_grad(include_variance=True).clone()
    if dp:
        g /= parallel.allreduce_flat(g)
    return g, float(loss)
half = slice(rank * N // 2, (rank + 1) * N // 2)
g_exact, l_exact = grad(half, True, True)
g_ddp, _ = grad(half, False, True)
g_big, l_big = grad(slice(0, N), False, False)
lsum = torch.tensor([l_exact], device="cuda"); dist.all_reduce(lsum)
if rank == 0:
    n = float(g_big.norm())
    print(json.dumps(dict(rel_exact=float((g_exact - g_big).norm()) / n, rel_ddp=float((g_ddp - g_big).norm()) / n,
                          loss_mean=float(lsum) / world, loss_big=l_big)))
dist.destroy_process_group()
'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        script = os.path.join(td, "w.py")
        open(script, "w").write(code)
        env = dict(os.environ, ES_REPO=REPO)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), script]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["rel_exact"] < 2e-5, d                      # the big batch's gradient to fp32 rounding
    assert abs(d["loss_mean"] - d["loss_big"]) < 1e-5 * max(1.0, abs(d["loss_big"])), d
    assert d["rel_ddp"] > 50 * d["rel_exact"], d          # the averaged per-rank ratios are a different (standard DDP) objective
