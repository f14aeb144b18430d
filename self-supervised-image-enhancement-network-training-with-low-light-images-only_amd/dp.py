"""Data-parallel host logic: one process per GPU, patches sharded across ranks, ONE all-reduce of the flat fp32
gradient buffer per step over RCCL/xGMI (backend "nccl" on ROCm), mean folded into the fused Adam kernel.

The reference is single-process, so this has no reference counterpart; the invariant it
must keep is "k-rank averaged gradients == 1-rank gradients on the concatenated batch" (every loss is a mean over
the batch, model.py:551-555, so with equal shards the global gradient is exactly the rank average).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


# When set (bench.py under torch.distributed.run), the gradient all-reduce is issued even at world size 1 so that a single-GPU
# launch exercises RCCL initialisation, the collective's stream ordering against the backward's side-stream join and the
# Adam scale exactly like an N-GPU launch does.
FORCE_COLLECTIVE = False


def init_from_env(backend: str | None = None):
    """(rank, world, local_rank); initialises torch.distributed from RANK/WORLD_SIZE/MASTER_* when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> range:
    """contiguous, equal shards (global batch must divide evenly so the rank-average equals the global mean)"""
    if n_items % world:
        raise ValueError(f"global batch {n_items} is not divisible by world size {world}")
    per = n_items // world
    return range(rank * per, (rank + 1) * per)


def rank_seed(seed: int, rank: int) -> int:
    return seed + rank


def allreduce_flat_(flat_grads: torch.Tensor, world: int) -> float:
    """sum-all-reduce the flat gradient buffer in place; returns the scale (1/world) the optimiser must apply"""
    if world > 1 or (FORCE_COLLECTIVE and dist.is_initialized()):
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return 1.0 / world


def broadcast_flat_(flat_params: torch.Tensor, world: int, src: int = 0):
    if world > 1:
        dist.broadcast(flat_params, src=src)
