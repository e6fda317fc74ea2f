#!/usr/bin/env python3
"""Round 6: the sincos rule of make_desc (every evaluation through the LDS table, HAMK_TRIG_LUT=1, or one table evaluation per step +
rotations about the step's midpoint, =2) re-measured for the systems whose right-hand sides the symbolic mass matrix made short.
RK4 steps/s of hamk_rk4_steps, 1000 fused steps, BASELINE sizes, same box.   python scripts/trig_rule_ab.py [--compile-only]"""
import json
import os
os.environ["HAMK_TEST_OVERRIDES"] = "1"
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
COMPILE_ONLY = "--compile-only" in sys.argv
from hamilton_amd import api, examples
if not COMPILE_ONLY:
    import torch

for name, B in (("threeBodyPolar", 1 << 18), ("spring", 1 << 20), ("room", 1 << 20)):
    spec = examples.get(name)
    for lut in ("1", "2"):
        os.environ["HAMK_TRIG_LUT"] = lut
        s = api.system_from_spec(spec)
        if COMPILE_ONLY:
            print(name, lut, s.code_size, flush=True)
            continue
        q, qd = examples.sample_config(spec, 0, B)
        ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
        st = api.Phase(ph.positions.clone(), ph.momenta.clone())
        api.rk4Steps(spec.dt, 1000, s, st, inplace=True)
        torch.cuda.synchronize()
        best = None
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); api.rk4Steps(spec.dt, 1000, s, st, inplace=True); e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        print(json.dumps({"what": "trig_rule_ab", "system": name, "B": B, "HAMK_TRIG_LUT": int(lut), "rk4_steps_per_s": B * 1000 / (best * 1e-3)}), flush=True)
