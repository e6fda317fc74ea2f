#!/usr/bin/env python3
"""Round 6: coordinate maps with a DENSE Jacobian on the four-lane kernels (hamk_quad.hpp assemble_dense: K accumulated in tiles,
compile-time seeds) against the wave-cooperative kernels they used to run on (every lane evaluates the whole tape at one-direction
jets), one box, RK4 steps/s of hamk_rk4_steps at B = 16 384 and 65 536, 20 fused steps per launch; hamEqs and one RK4 step of
both against the oracle on a sample.
  python scripts/dense_quad_ab.py [--compile-only] [names...] > gpurun_out/r06_dense_quad_ab.jsonl"""
import json
import os
os.environ["HAMK_TEST_OVERRIDES"] = "1"               # this script drives libhamk.so through its HAMK_* test overrides
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
COMPILE_ONLY = "--compile-only" in sys.argv
NAMES = [a for a in sys.argv[1:] if not a.startswith("--")] or ["dense18", "dense24", "dense32"]
import numpy as np
from hamilton_amd import _abi, api, examples

if not COMPILE_ONLY:
    import torch
    from oracle import oracle


def rate(s, spec, B, nsteps=20):
    q, qd = examples.sample_config(spec, 0, B)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    st = api.Phase(ph.positions.clone(), ph.momenta.clone())
    api.rk4Steps(spec.dt, nsteps, s, st, inplace=True)
    torch.cuda.synchronize()
    best = None
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); api.rk4Steps(spec.dt, nsteps, s, st, inplace=True); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return B * nsteps / (best * 1e-3), int(torch.count_nonzero(s.last_status))


for name in NAMES:
    spec = examples.get(name)
    for tag, mapping in (("quad-dense", _abi.MAP_QUAD), ("wave", _abi.MAP_WAVE)):
        t0 = time.time()
        s = api.system_from_spec(spec, {"mapping": mapping})
        build_s = time.time() - t0
        info = [l for l in s.build_info.splitlines() if l.startswith(("hamk_rk4_steps_k", "hamk_hameqs_k"))]
        if COMPILE_ONLY:
            print(name, tag, f"{build_s:.0f} s", info, flush=True)
            continue
        o = oracle.OracleSystem(spec)
        S = 64
        q, qd = examples.sample_config(spec, 7, S)
        p = o.to_phase_batch(q, qd)
        tq, tp = torch.from_numpy(q).cuda(), torch.from_numpy(p).cuda()
        dq, dp = api.hamEqs(s, api.Phase(tq, tp))
        odq, odp, _ = o.hameqs_batch(q, p)
        rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))
        e_h = max(rel(dq.cpu().numpy(), odq), rel(dp.cpu().numpy(), odp))
        one = api.rk4Steps(spec.dt, 1, s, api.Phase(tq, tp))
        oq, op = o.rk4_steps_batch(q, p, spec.dt, 1)
        e_1 = max(rel(one.positions.cpu().numpy(), oq), rel(one.momenta.cpu().numpy(), op))
        for B in (16384, 65536):
            r, flagged = rate(s, spec, B)
            print(json.dumps({"what": "dense_quad_ab", "system": name, "n": spec.n, "m": spec.m, "mapping": tag, "B": B, "rk4_steps_per_s": r, "flagged": flagged,
                              "hameqs_rel_err_vs_oracle": e_h, "one_step_rel_err_vs_oracle": e_1, "build_s_or_cache": round(build_s, 1), "kernels": info}), flush=True)
