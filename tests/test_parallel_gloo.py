"""CPU, world_size 2 (gloo): the data-parallel gradient path — one flat all-reduce averages gradients; parameter broadcast."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from endosurf_amd import parallel
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(rank)
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(()))]
    parallel.broadcast_parameters(params, src=0)
    sums = [float(p.detach().sum()) for p in params]
    for i, p in enumerate(params[:2]):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    params[2].grad = None                                   # missing gradient counts as zero
    parallel.allreduce_gradients(params)
    q.put((rank, sums, [float(p.grad.reshape(-1)[0]) for p in params]))
    dist.destroy_process_group()


def test_allreduce_and_broadcast_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, g0), (_, s1, g1) = res
    assert s0 == s1                                          # broadcast made the replicas identical
    assert g0 == g1 == [1.5, 3.0, 0.0]                       # mean over ranks of (1,2)->1.5, (2,4)->3, missing->0


def test_flat_bucket_roundtrip():
    from endosurf_amd import parallel
    ps = [torch.nn.Parameter(torch.zeros(2, 3)), torch.nn.Parameter(torch.zeros(4))]
    ps[0].grad = torch.arange(6.0).view(2, 3)
    flat = parallel.flatten_grads(ps)
    assert flat.tolist() == [0, 1, 2, 3, 4, 5, 0, 0, 0, 0]
    parallel.unflatten_to_grads(flat + 1, ps)
    assert ps[0].grad.tolist() == [[1, 2, 3], [4, 5, 6]] and ps[1].grad.tolist() == [1, 1, 1, 1]


def _flat_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch
    from endosurf_amd import parallel
    parallel.init_distributed("gloo")
    g = torch.full((1000,), float(rank + 1))
    w = parallel.allreduce_flat(g)
    q.put((rank, w, float(g[0]), float(g[-1])))
    torch.distributed.destroy_process_group()


def test_allreduce_flat_sums_one_bucket():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + 7
    procs = [ctx.Process(target=_flat_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, 2, 3.0, 3.0), (1, 2, 3.0, 3.0)]


def _gather_worker(rank, world, port, q, H, W):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from endosurf_amd import parallel
    parallel.init_distributed("gloo")
    r0, rows = parallel.frame_rows(H, rank, world)
    # pixel (y, x) carries the value 1000 y + x in every channel (+ channel index): the assembled frame is checked entry by entry
    yy, xx = torch.meshgrid(torch.arange(r0, r0 + rows, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    base = (1000.0 * yy + xx).reshape(-1, 1)
    parts = {"color": base + torch.arange(3.0), "depth": base + 10.0, "normal": base + 20.0 + torch.arange(3.0)}
    out = parallel.gather_frame(parts, H, W)
    if rank == 0:
        q.put({k: v.numpy() for k, v in out.items()})
    else:
        assert out is None
        q.put(None)
    dist.destroy_process_group()


def test_gather_frame_world2_uneven_rows():
    """cfg5 on N ranks: rank 0 ends with ONE [H, W, C] image per output, assembled from the ranks' row slabs (H odd: the slabs differ)."""
    import numpy as np
    H, W = 7, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q, H, W)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out = [r for r in res if r is not None]
    assert len(out) == 1
    out = out[0]
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    base = 1000.0 * yy + xx
    assert out["color"].shape == (H, W, 3) and out["depth"].shape == (H, W, 1) and out["normal"].shape == (H, W, 3)
    for c in range(3):
        assert np.array_equal(out["color"][..., c], base + c) and np.array_equal(out["normal"][..., c], base + 20 + c)
    assert np.array_equal(out["depth"][..., 0], base + 10)


def test_gather_frame_single_process_and_row_split():
    from endosurf_amd import parallel
    assert [parallel.frame_rows(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 64)]
    assert [parallel.frame_rows(7, r, 4) for r in range(4)] == [(0, 2), (2, 2), (4, 2), (6, 1)]
    assert parallel.frame_rows(3, 3, 4) == (3, 0)
    parts = {"color": torch.arange(24.0).reshape(8, 3), "depth": torch.arange(8.0).reshape(8, 1)}
    out = parallel.gather_frame(parts, 2, 4)
    assert out["color"].shape == (2, 4, 3) and torch.equal(out["color"].reshape(8, 3), parts["color"])
    assert torch.equal(out["depth"].reshape(8, 1), parts["depth"])


def test_local_device_fails_fast_instead_of_folding(monkeypatch):
    """A rank without a GPU of its own must raise with the numbers in the message (bench.py folded 8 ranks onto fewer GPUs)."""
    import pytest
    from endosurf_amd import parallel
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    assert parallel.local_device(1, 8) == 1
    with pytest.raises(RuntimeError, match="only 2 GPU"):
        parallel.local_device(2, 8)
