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


# ---------------------------------------------------------------------------------------
# ensemble checkpoint / resume.  The reference's "resume" story is
# `iterate (stepHam dt s)` on a `Phase n` value (README.md:150); an ensemble is two flat fp64
# arrays plus what is needed to regenerate or extend it.
# ---------------------------------------------------------------------------------------
CHECKPOINT_VERSION = 1


def save_checkpoint(path: str, system: str, q, p, *, t: float, step: int, dt: float, seed: int,
                    first_index: int = 0, status=None) -> None:
    """Flat binary (.npz, uncompressed) dump of an SoA ensemble shard."""
    def host(a):
        return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    q, p = host(q), host(p)
    if q.shape != p.shape or q.ndim != 2:
        raise ValueError("q and p must both be [n, B]")
    extra = {} if status is None else {"status": host(status).astype(np.int32)}
    np.savez(path, version=np.int64(CHECKPOINT_VERSION), system=np.array(system), q=q.astype(np.float64),
             p=p.astype(np.float64), t=np.float64(t), step=np.int64(step), dt=np.float64(dt),
             seed=np.int64(seed), first_index=np.int64(first_index), **extra)


def load_checkpoint(path: str) -> dict:
    with np.load(path if path.endswith(".npz") else path + ".npz", allow_pickle=False) as z:
        if int(z["version"]) != CHECKPOINT_VERSION:
            raise ValueError(f"unsupported checkpoint version {int(z['version'])}")
        out = {k: z[k] for k in z.files}
    out["system"] = str(out["system"])
    for k in ("t", "dt"):
        out[k] = float(out[k])
    for k in ("step", "seed", "first_index", "version"):
        out[k] = int(out[k])
    return out
