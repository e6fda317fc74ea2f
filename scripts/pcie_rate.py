#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer path for large ensembles (never bench.py's `value`): one RK4 step of the config-2
ensemble handed over as numpy arrays, 2 x 2 x 8 B x B in and out per call (hipMemcpyAsync between the caller's pageable
memory and device staging), IN PLACE -- the caller's arrays are reused -- and into fresh result arrays, where the first
touch of every result page is what the call spends its time on:
  python scripts/pcie_rate.py > gpurun_out/r03_pcie_rate.jsonl
(A pinned double-buffered bounce path with a multi-threaded memcpy was built and measured against this: 25-41 GB/s
against 50-52 -- the runtime's own pageable path is the faster one on these boxes, and was removed again.)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one():
    sys.path.insert(0, ROOT)
    from hamilton_amd import api, examples
    spec = examples.get("doublePendulum")
    s = api.system_from_spec(spec)
    for B in (1 << 20, 1 << 22):
        q, qd = examples.sample_config(spec, 0, B)
        ph = api.toPhase(s, api.Config(q, qd))
        for inplace, nsteps in ((True, 1), (False, 1), (True, 10), (True, 100), (True, 1000)):
            api.rk4Steps(0.01, nsteps, s, ph, inplace=inplace)
            best = None
            for _ in range(5):
                t0 = time.perf_counter()
                api.rk4Steps(0.01, nsteps, s, ph, inplace=inplace)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            moved = 2 * 2 * spec.n * B * 8
            print(json.dumps({"B": B, "in_place": inplace, "fused_rk4_steps_per_call": nsteps, "call_ms": best * 1e3, "bytes_each_way": moved // 2,
                              "GB_per_s_both_ways": moved / best / 1e9, "trajectory_steps_per_s": B * nsteps / best}), flush=True)


if __name__ == "__main__":
    one()
