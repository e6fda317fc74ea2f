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


def pin_for_ensemble(system, total: int):
    """Every shard of a `total`-member ensemble on the mapping the library picks for the WHOLE ensemble
    (hamk_options::ensemble_size): with the choice left per launch, a shard of 8 192 of chain16's 65 536 would run the
    four-lane kernels where the single-GPU run uses the lane kernels -- equal to roundoff, not bitwise.  Call once per
    handle before the first launch of a sharded or resumed run; returns the handle."""
    return system.set_ensemble_size(int(total))


def gather_state(q, p, dist, world: int):
    """All-gather equally sized shards of (q, p) [n, B] -> [n, world*B] on every rank."""
    import torch
    mine = torch.stack([q, p]).contiguous()
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    full = torch.cat(parts, dim=2)              # [2, n, world*B], rank-major = global index order
    return full[0], full[1]


# ---------------------------------------------------------------------------------------
# ensemble checkpoint / resume.  The reference's "resume" story is
# `iterate (stepHam dt s)` on a `Phase n` value (README.md:150); an ensemble is two flat fp64
# arrays plus what is needed to regenerate or extend it.  ONE format: the C ABI's
# (hamk_checkpoint_write / _info / _read: 64-byte header, SoA state, SHA-256); a sharded run
# writes one file per rank, named by (rank, world) -- the shard's first global index follows
# from shard_bounds / weak_bounds, the per-index seed is in the header.
# ---------------------------------------------------------------------------------------
def shard_path(base: str, rank: int, world: int) -> str:
    return f"{base}.{rank:04d}-of-{world:04d}.hamkckp"


def save_shard(base: str, rank: int, world: int, phase, n: int, *, steps_done: int, seed: int, t: float) -> str:
    """This rank's shard (numpy arrays or torch CUDA tensors, [n, B]) through hamk_checkpoint_write."""
    from . import api
    path = shard_path(base, rank, world)
    api.saveCheckpoint(path, phase, n, steps_done=steps_done, seed=seed, t=t)
    return path


def load_shard(base: str, rank: int, world: int, device=None):
    """(Phase, info) of this rank's shard; device=None: numpy arrays, else CUDA tensors on it."""
    from . import api
    return api.loadCheckpoint(shard_path(base, rank, world), device)
