"""Batch sharding across the GPUs of one node.

Every ciphertext pair is independent, so the hot path shards with NO data-path collective:
pair p of a global batch of B goes to rank p*G//B (contiguous ranges, deterministic, so the
per-pair results are identical for any GPU count).  The only exchange is an all-gather of the
per-pair result TERM COUNTS (one int64 per pair) -- RCCL over xGMI with backend "nccl" on the
GPUs, gloo in the CPU tests.  Nothing here touches ciphertext words.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the pairs owned by `rank`: pair p belongs to rank p*world//total_pairs."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    lo = -(-rank * total_pairs // world)          # ceil(rank*B/G)
    hi = -(-(rank + 1) * total_pairs // world)
    return lo, min(hi, total_pairs)


def owner_of(pair: int, total_pairs: int, world: int) -> int:
    return pair * world // total_pairs


def gather_term_counts(local_counts: torch.Tensor, total_pairs: int,
                       group: Optional[dist.ProcessGroup] = None,
                       out: Optional[torch.Tensor] = None, force: bool = False) -> torch.Tensor:
    """All-gather the per-pair result term counts of every rank into one tensor of
    `total_pairs` int64, in global pair order.  Equal shards use one
    all_gather_into_tensor; uneven shards are padded to the largest shard first."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return local_counts
    world = dist.get_world_size(group)
    sizes = [shard_range(total_pairs, r, world) for r in range(world)]
    lens = [hi - lo for lo, hi in sizes]
    rank = dist.get_rank(group)
    assert local_counts.numel() == lens[rank], (local_counts.numel(), lens[rank])
    if len(set(lens)) == 1:
        if out is None:
            out = torch.empty(total_pairs, dtype=local_counts.dtype, device=local_counts.device)
        dist.all_gather_into_tensor(out, local_counts.contiguous(), group=group)
        return out
    width = max(lens)
    padded = torch.zeros(width, dtype=local_counts.dtype, device=local_counts.device)
    padded[: lens[rank]] = local_counts
    buf = torch.empty(world * width, dtype=local_counts.dtype, device=local_counts.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts: List[torch.Tensor] = [buf[r * width: r * width + lens[r]] for r in range(world)]
    return torch.cat(parts)


def product_term_counts(off_left: torch.Tensor, off_right: torch.Tensor) -> torch.Tensor:
    """t1_b * t2_b from CSR term offsets (src/Ciphertext.cpp:146: newlen/dL).  Pure
    metadata; on the GPU path the same numbers come out of csgn_mul_ragged_plan."""
    return (off_left[1:] - off_left[:-1]) * (off_right[1:] - off_right[:-1])
