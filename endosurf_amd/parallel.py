"""Data parallelism over ray batches: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests), weights replicated, each rank draws its own ray batch, ONE all-reduce of the flat fp32 gradient
bucket per step (1 654 951 floats = 6.6 MB).  Nothing else in the hot path communicates: rays are independent."""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialise from torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT). Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


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


def allreduce_flat(flat: torch.Tensor, group=None) -> int:
    """Sum a flat gradient bucket across ranks in place (one collective); returns the world size (1 if not distributed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 1
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return dist.get_world_size(group)


def broadcast_parameters(params: Iterable[torch.Tensor], src: int = 0, group=None):
    """Make every rank start from rank ``src``'s weights (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    params = list(params)
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for p in params:
            p.copy_(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
