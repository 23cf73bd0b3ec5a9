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
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--mode", mode]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]              # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    if mode == "train":
        assert d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2" and d["roofline"]["frac"] > 0


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
