"""Data-parallel sharding of the crop-clip stream: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU tests of the host logic).

The path shards over independent units (every crop-clip is an independent backbone forward,
/root/reference/extract_features.py:85-89), so the only exchange is an all-gather of the per-clip
2048-d feature rows -- rank r owns rows [r*B_local, (r+1)*B_local) of every global batch, and
concatenation in rank order reproduces the reference's (n_clips, 10, 2048) ordering
(extract_features.py:93-100).  The reference itself has no distributed code.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process -> (0,0,1))."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced block of rank `rank` out of n units (first n % world ranks get one more)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def padded_local_rows(n: int, world: int) -> int:
    """Rows every rank contributes to the all-gather (the last ranks pad)."""
    return -(-n // world)


def all_gather_rows(local: torch.Tensor, group=None) -> torch.Tensor:
    """(B_local, C) on every rank -> (world*B_local, C), rank-major.  One collective, no copies
    besides RCCL's own (all_gather_into_tensor writes the output in place)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    if dist.get_backend(group) == "gloo":  # CPU tests / single-GPU rehearsal: gloo has no flat all-gather
        dist.all_gather(list(out.chunk(world, dim=0)), local.contiguous(), group=group)
    else:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def sharded_map_rows(fn: Callable[[torch.Tensor], torch.Tensor], units: torch.Tensor, group=None) -> torch.Tensor:
    """Apply `fn` (unit batch -> one feature row per unit) to this rank's contiguous block of a
    GLOBAL batch of independent units and return the rows of the whole batch, in order, on
    every rank.  Ragged splits are padded by repeating the block's last unit and trimmed after
    the gather, so every rank issues the same collective."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return fn(units)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = units.shape[0]
    lo, hi = shard_bounds(n, world, rank)
    rows = padded_local_rows(n, world)
    mine = units[lo:hi]
    if hi - lo < rows:  # pad (possibly from an empty block: reuse any unit of the batch)
        filler = (mine[-1:] if hi > lo else units[:1]).expand(rows - (hi - lo), *units.shape[1:])
        mine = torch.cat([mine, filler], dim=0)
    out = fn(mine)
    gathered = all_gather_rows(out.reshape(rows, -1), group)
    keep: List[torch.Tensor] = []
    for r in range(world):
        l, h = shard_bounds(n, world, r)
        keep.append(gathered[r * rows : r * rows + (h - l)])
    return torch.cat(keep, dim=0)
