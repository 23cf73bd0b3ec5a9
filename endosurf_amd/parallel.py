"""Data parallelism over ray batches: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests), weights replicated, each rank draws its own ray batch, ONE all-reduce of the flat fp32 gradient
bucket per step (1 654 951 floats = 6.6 MB).  Nothing else in the hot path communicates: rays are independent."""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None, force: bool = False) -> tuple:
    """Initialise from torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT). Returns (rank, world, local).
    ``force``: create the process group even for a single rank (drives the RCCL path on one GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_device(local, world))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def local_device(local: int, world: int) -> int:
    """GPU index of this rank: one rank per GPU.  Fails with a clear message when the node has fewer GPUs than local ranks instead
    of silently folding several ranks onto one device (RCCL would then hang or run all ranks on one GPU)."""
    n = torch.cuda.device_count()
    if local >= n:
        raise RuntimeError(f"local rank {local} of a {world}-rank job needs its own GPU, but only {n} GPU(s) are visible on this node "
                           f"(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = {os.environ.get('HIP_VISIBLE_DEVICES')!r} / "
                           f"{os.environ.get('ROCR_VISIBLE_DEVICES')!r}); launch at most {n} ranks per node")
    return local


def pin_rank_to_cores(local: int, local_world: int, reserve: int = 0) -> dict:
    """Give every local rank its own, DISJOINT set of host cores (os.sched_setaffinity) out of the cores this process may use at all
    (os.sched_getaffinity: the container's cpuset / the launcher's mask is honoured, never widened).  A training step is ~170 launches
    issued by one Python thread in 3-4 ms of host time; eight such threads plus their RCCL proxy threads migrating over one socket's
    cores is the first thing that goes wrong on a shared host.  The allowed cores are split into ``local_world`` contiguous runs in
    core-id order (neighbouring ids share a NUMA node / L3 on the usual enumeration).  Fewer allowed cores than ranks: nothing is
    pinned (and the return value says so).  -> {"pinned": bool, "cores": n, "first": id, "last": id, "allowed": n_allowed}"""
    if not hasattr(os, "sched_getaffinity"):
        return dict(pinned=False, reason="no sched_getaffinity on this platform")
    allowed = sorted(os.sched_getaffinity(0))
    n = len(allowed)
    if local_world <= 1 or n < local_world:
        return dict(pinned=False, allowed=n, reason="one rank" if local_world <= 1 else f"{n} allowed cores for {local_world} ranks")
    per = n // local_world
    mine = allowed[local * per:(local + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError as e:      # (a sandbox may forbid it: run unpinned rather than fail)
        return dict(pinned=False, allowed=n, reason=f"sched_setaffinity: {e}")
    torch.set_num_threads(max(1, min(4, per - reserve)))      # the host side of a step is one launch thread; keep torch's pool small
    return dict(pinned=True, cores=len(mine), first=mine[0], last=mine[-1], allowed=n)


def flatten_grads(params: List[torch.Tensor]) -> torch.Tensor:
    """One contiguous fp32 bucket holding every gradient (missing gradients count as zero)."""
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])


def unflatten_to_grads(flat: torch.Tensor, params: List[torch.Tensor]):
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view(p.shape)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n


def allreduce_gradients(params: Iterable[torch.Tensor], group=None):
    """Average gradients across ranks with a single all-reduce of the flat bucket (standard DDP semantics)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    params = list(params)
    flat = flatten_grads(params)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(dist.get_world_size(group))
    unflatten_to_grads(flat, params)


def allreduce_flat(flat: torch.Tensor, group=None, force: bool = False) -> int:
    """Sum a flat gradient bucket across ranks in place (one collective); returns the world size (1 if not distributed).
    ``force``: issue the collective even in a one-rank group (smoke test of the RCCL path on a single GPU)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return 1
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return world


def broadcast_parameters(params: Iterable[torch.Tensor], src: int = 0, group=None):
    """Make every rank start from rank ``src``'s weights (one flat broadcast; also issued in a one-rank group, where it is the
    identity: the first collective of a job doubles as the check that the backend works)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    params = list(params)
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for p in params:
            p.copy_(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()


def gather_frame(parts: dict, rows_total: int, width: int, dst: int = 0, group=None):
    """Assemble ONE image per output from the row slabs the ranks rendered (cfg5: ``bench.py --config 5 --gpus N``, the end of the
    reference's eval loop, trainer_endosurf.py:221-240, which concatenates ray chunks into one frame).

    ``parts``: {name: tensor [rows_local * width, C]} of this rank's consecutive rows (rank r holds rows [r * ceil(H / world), ...));
    all outputs travel in ONE all-gather of a packed [rows_max * width, sum C] buffer (short slabs are padded).  Returns
    {name: [rows_total, width, C]} on rank ``dst`` and None elsewhere; without a process group the local slab IS the frame."""
    names = sorted(parts)
    chans = [parts[k].shape[-1] for k in names]
    packed = torch.cat([parts[k].reshape(-1, c).to(torch.float32) for k, c in zip(names, chans)], dim=-1).contiguous()
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    world = dist.get_world_size(group) if distributed else 1
    rows_max = -(-rows_total // world)
    if packed.shape[0] > rows_max * width or packed.shape[0] % width:
        raise ValueError(f"slab of {packed.shape[0]} rays does not fit {rows_max} rows x {width}")
    if distributed:
        pad = rows_max * width - packed.shape[0]
        if pad:
            packed = torch.cat([packed, packed.new_zeros(pad, packed.shape[1])], 0)
        slabs = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(slabs, packed, group=group)
        if dist.get_rank(group) != dst:
            return None
        full = torch.cat(slabs, 0)[:rows_total * width]
    else:
        if packed.shape[0] != rows_total * width:
            raise ValueError(f"single process: expected the whole frame ({rows_total * width} rays), got {packed.shape[0]}")
        full = packed
    out, c0 = {}, 0
    for k, c in zip(names, chans):
        out[k] = full[:, c0:c0 + c].reshape(rows_total, width, c)
        c0 += c
    return out


def frame_rows(rows_total: int, rank: int, world: int):
    """(first row, row count) of rank's slab: ceil(H / world) rows per rank, the last ranks may hold fewer (or none)."""
    per = -(-rows_total // world)
    r0 = min(rank * per, rows_total)
    return r0, min(per, rows_total - r0)
