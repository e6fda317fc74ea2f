"""CPU, world_size 2, gloo: the N>1 path -- shard by global index, no data-path collective,
one final gather -- reproduces the single-process ensemble bit for bit.  The compute
stand-in on CPU is the oracle (allowed in tests); on GPUs bench.py runs the HIP path with
backend nccl (= RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from hamilton_amd import ensemble
from hamilton_amd import examples as E


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, per_rank, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    spec = E.get("doublePendulum")
    o = oracle.OracleSystem(spec)
    lo, hi = ensemble.weak_bounds(per_rank, rank)
    q, qd = E.sample_config(spec, lo, hi - lo)
    p = o.to_phase_batch(q, qd, threads=1)
    q1, p1 = o.rk4_steps_batch(q, p, 0.01, 5, threads=1)
    gq, gp = ensemble.gather_state(torch.from_numpy(q1), torch.from_numpy(p1), dist, world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)            # the bench's max-over-ranks timing reduction
    assert float(t[0]) == world
    if rank == 0:
        np.save(out_path, np.stack([gq.numpy(), gp.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_reproduce_single_process(tmp_path, oracle_lib):
    world, per_rank = 2, 96
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), per_rank, out), nprocs=world, join=True)
    got = np.load(out)
    spec = E.get("doublePendulum")
    o = oracle_lib.OracleSystem(spec)
    q, qd = E.sample_config(spec, 0, world * per_rank)
    p = o.to_phase_batch(q, qd, threads=1)
    q1, p1 = o.rk4_steps_batch(q, p, 0.01, 5, threads=1)
    np.testing.assert_array_equal(got[0], q1)
    np.testing.assert_array_equal(got[1], p1)


def test_shard_bounds_partition():
    for total in (0, 1, 7, 262144, 1000003):
        for world in (1, 2, 3, 8):
            spans = [ensemble.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        ensemble.shard_bounds(10, 2, 2)


def test_checkpoint_roundtrip(tmp_path, oracle_lib, hamk_lib):
    """Resume = reload the two flat arrays and keep stepping: identical bits to an uninterrupted run.  One checkpoint
    format -- the C ABI's (hamk_checkpoint_*), one file per rank of a sharded run."""
    from hamilton_amd import api
    spec = E.get("spring")
    o = oracle_lib.OracleSystem(spec)
    world = 2
    q, qd = E.sample_config(spec, 0, 64)
    p = o.to_phase_batch(q, qd, threads=1)
    qf, pf = o.rk4_steps_batch(q, p, spec.dt, 7, threads=1)
    for rank in range(world):
        lo, hi = ensemble.shard_bounds(64, world, rank)
        q1, p1 = o.rk4_steps_batch(q[:, lo:hi], p[:, lo:hi], spec.dt, 4, threads=1)
        ensemble.save_shard(str(tmp_path / "ens"), rank, world, api.Phase(q1, p1), spec.n, steps_done=4, seed=E.SEED, t=4 * spec.dt)
    for rank in range(world):
        lo, hi = ensemble.shard_bounds(64, world, rank)
        ph, info = ensemble.load_shard(str(tmp_path / "ens"), rank, world)
        assert info == {"n": spec.n, "B": hi - lo, "steps_done": 4, "seed": E.SEED, "t": 4 * spec.dt}
        q2, p2 = o.rk4_steps_batch(ph.positions, ph.momenta, spec.dt, 3, threads=1)
        np.testing.assert_array_equal(q2, qf[:, lo:hi])
        np.testing.assert_array_equal(p2, pf[:, lo:hi])
    # a header that does not describe its file is rejected before anything is allocated from it
    path = ensemble.shard_path(str(tmp_path / "ens"), 0, world)
    raw = bytearray(open(path, "rb").read())
    raw[16:24] = (1 << 50).to_bytes(8, "little")              # B
    open(path, "wb").write(bytes(raw))
    with pytest.raises(api.HamkError):
        api.checkpointInfo(path)
    with pytest.raises(api.HamkError):
        api.loadCheckpoint(path)
