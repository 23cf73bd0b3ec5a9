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
