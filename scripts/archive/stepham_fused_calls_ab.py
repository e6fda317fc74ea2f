#!/usr/bin/env python3
"""stepHam calls/s as a function of the calls fused into one launch (hamk_step_ham_iterate): how much of a one-call launch is
launch / ramp-up / tail.  python scripts/archive/stepham_fused_calls_ab.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from hamilton_amd import api, examples
for name, B in (("doublePendulum", 1 << 20), ("threeBodyPolar", 262144), ("chain6", 262144), ("chain8", 65536), ("chain8", 262144), ("chain12", 65536), ("chain16", 65536), ("chain16", 131072)):
    spec = examples.get(name)
    s = api.system_from_spec(spec)
    q, qd = examples.sample_config(spec, 0, B)
    if name.startswith("chain"):
        qd = 0.3 * np.cos(np.arange(spec.n * B).reshape(spec.n, B) * 0.7)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    for K in (1, 2, 8, 32):
        st = api.Phase(ph.positions.clone(), ph.momenta.clone())
        f = (lambda: api.stepHam(spec.dt, s, st)) if K == 1 else (lambda: api.iterateStepHam(spec.dt, K, s, st))
        f(); torch.cuda.synchronize()
        best = None
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1); best = ms if best is None else min(best, ms)
        print(json.dumps({"system": name, "B": B, "calls_per_launch": K, "us_per_launch": best * 1e3, "calls_per_s": B * K / (best * 1e-3)}), flush=True)
