"""Ensemble sharding across GPUs: the only parallelism the path has.

`hamEqs` is a pure function of one `Phase` (Hamilton.hs:370-387): trajectories never
exchange data, so an ensemble shards by contiguous global index range with NO data-path
collective.  Initial conditions come from a per-index counter RNG (examples.sample_config),
so any shard layout reproduces bit-identical inputs.  The single collective is the final
gather of the state (RCCL all_gather over xGMI on GPUs; gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Strong sharding: contiguous [lo, hi) of `total` trajectories for `rank`; remainders go
    to the lowest ranks so shard sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def weak_bounds(per_rank: int, rank: int) -> Tuple[int, int]:
    """Weak sharding (bench.py): every rank owns `per_rank` trajectories."""
    return rank * per_rank, (rank + 1) * per_rank


def gather_state(q, p, dist, world: int):
    """All-gather equally sized shards of (q, p) [n, B] -> [n, world*B] on every rank."""
    import torch
    mine = torch.stack([q, p]).contiguous()
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    full = torch.cat(parts, dim=2)              # [2, n, world*B], rank-major = global index order
    return full[0], full[1]
