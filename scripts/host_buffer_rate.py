#!/usr/bin/env python
"""The C ABI takes caller-owned HOST or DEVICE arrays (include/hamk.h, hamk_rk4_steps / hamk_rkf45_steps).  bench.py's `value`
is quoted with the state resident in HBM; this script measures the same launches handed HOST arrays (numpy: staged over PCIe by
libhamk.so, in and out, inside the timed call) next to device-resident ones, so that DESIGN.md can state the PCIe-inclusive
rate.  One JSON line per (system, steps per call).

    HAMK_TEST_OVERRIDES is not needed; run on a GPU box:  python scripts/host_buffer_rate.py > gpurun_out/host_buffer_rate.jsonl
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--systems", default="doublePendulum:1048576:0.01,threeBodyPolar:262144:0.002,chain16:65536:0.005")
    ap.add_argument("--steps", default="1,10,100,1000")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()

    import torch
    from hamilton_amd import api, examples

    dev = torch.device("cuda:0")
    for item in a.systems.split(","):
        name, B, dt = item.split(":")
        B, dt = int(B), float(dt)
        spec = examples.get(name)
        s = api.system_from_spec(spec)
        n = s.n
        cfg = api.sampleConfig(s, spec.q_box, spec.qd_box, 0, B, examples.SEED, dev)
        ph = api.toPhase(s, cfg)
        qd, pd = ph.positions.contiguous(), ph.momenta.contiguous()
        qh, ph_h = qd.cpu().numpy().copy(), pd.cpu().numpy().copy()
        for nsteps in [int(x) for x in a.steps.split(",")]:
            if nsteps * B > 1.2e9 and name != "doublePendulum":
                continue

            def run_dev():
                q, p = qd.clone(), pd.clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                api.rk4Steps(dt, nsteps, s, api.Phase(q, p), inplace=True)
                s.synchronize()
                torch.cuda.synchronize()
                return time.perf_counter() - t0

            def run_host():
                q, p = qh.copy(), ph_h.copy()
                t0 = time.perf_counter()
                api.rk4Steps(dt, nsteps, s, api.Phase(q, p), inplace=True)          # returns with the results in q, p
                return time.perf_counter() - t0, q

            run_dev(); run_host()
            td = min(run_dev() for _ in range(a.reps))
            th, qout = min((run_host() for _ in range(a.reps)), key=lambda r: r[0])
            # the two paths run the same kernel on the same numbers
            q, p = qd.clone(), pd.clone()
            api.rk4Steps(dt, nsteps, s, api.Phase(q, p), inplace=True)
            same = bool(np.array_equal(q.cpu().numpy(), qout))
            bytes_moved = 2 * 2 * n * B * 8                                       # q, p in and out
            print(json.dumps({"system": name, "n": n, "B": B, "steps_per_call": nsteps,
                              "device_resident_steps_per_s": B * nsteps / td, "host_arrays_steps_per_s": B * nsteps / th,
                              "ms_device": td * 1e3, "ms_host": th * 1e3, "state_bytes_over_pcie": bytes_moved,
                              "staging_GBps": bytes_moved / max(th - td, 1e-9) / 1e9, "bitwise_equal": same}), flush=True)


if __name__ == "__main__":
    main()
