#!/usr/bin/env python3
"""sincos_lut's nine fp64 literals in VECTOR registers (-DHAMK_TRIG_CONST_VGPR=1) instead of re-materialised SGPR pairs:
stepHam calls/s (8 calls per launch) and RK4 steps/s.  python scripts/archive/trig_const_vgpr_ab.py [--compile-only]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
COMPILE_ONLY = "--compile-only" in sys.argv
import numpy as np
from hamilton_amd import _abi, api, examples
if not COMPILE_ONLY:
    import torch
def best_ms(f, reps=4):
    f(); torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1); best = ms if best is None else min(best, ms)
    return best
SYS = [("doublePendulum", 1 << 20, 400), ("spring", 1 << 20, 200), ("threeBodyPolar", 262144, 200), ("chain6", 262144, 100), ("chain8", 65536, 100), ("chain10", 65536, 50),
       ("chain12", 65536, 50), ("chain14", 65536, 20), ("chain16", 65536, 20)]
for name, B, nrk in SYS:
    spec = examples.get(name)
    for tag, flags in (("literals", ""), ("vgpr", "-DHAMK_TRIG_CONST_VGPR=1")):
        os.environ["HAMK_HIPRTC_FLAGS"] = flags
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE})
        if COMPILE_ONLY:
            print(name, tag, [l.split()[0] + " " + l.split()[1] + " " + l.split()[-1] for l in s.build_info.splitlines() if "rkf45" in l or "rk4" in l], flush=True); continue
        q, qd = examples.sample_config(spec, 0, B)
        if name.startswith("chain"):
            qd = 0.3 * np.cos(np.arange(spec.n * B).reshape(spec.n, B) * 0.7)
        ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
        st = api.Phase(ph.positions.clone(), ph.momenta.clone())
        ms = best_ms(lambda: api.iterateStepHam(spec.dt, 8, s, st))
        st2 = api.Phase(ph.positions.clone(), ph.momenta.clone())
        ms2 = best_ms(lambda: api.rk4Steps(spec.dt, nrk, s, st2, inplace=True))
        print(json.dumps({"system": name, "variant": tag, "stepham_calls_per_s": B * 8 / (ms * 1e-3), "rk4_steps_per_s": B * nrk / (ms2 * 1e-3)}), flush=True)
