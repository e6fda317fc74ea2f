#!/usr/bin/env python3
"""Why does the parked adaptive stepper issue at ~0.37 at ONE wavefront per SIMD (n = 8..12) when the RK4 kernel of the same
system reaches 0.64?  Variants: no scheduling fences around the right-hand side; sincos without the LDS table.
python scripts/archive/rkf_one_wave_probe.py [--compile-only]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
COMPILE_ONLY = "--compile-only" in sys.argv
import numpy as np
from hamilton_amd import _abi, api, examples
if not COMPILE_ONLY:
    import torch
def rate(s, spec, B, dt):
    q, qd = examples.sample_config(spec, 0, B)
    qd = 0.3 * np.cos(np.arange(spec.n * B).reshape(spec.n, B) * 0.7)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    st = api.Phase(ph.positions.clone(), ph.momenta.clone())
    K = 8
    api.iterateStepHam(dt, K, s, st); torch.cuda.synchronize()
    best = None
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); api.iterateStepHam(dt, K, s, st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1); best = ms if best is None else min(best, ms)
    return B * K / (best * 1e-3)
for name in ("chain8", "chain10", "chain12"):
    spec = examples.get(name)
    for tag, flags, lut in (("shipped", "", None), ("nofence", "-DHAMK_RKF_NO_FENCE", None), ("notable", "", "0"), ("nofence-notable", "-DHAMK_RKF_NO_FENCE", "0")):
        os.environ["HAMK_HIPRTC_FLAGS"] = flags
        if lut is None: os.environ.pop("HAMK_TRIG_LUT", None)
        else: os.environ["HAMK_TRIG_LUT"] = lut
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE})
        if COMPILE_ONLY:
            print(name, tag, [l for l in s.build_info.splitlines() if "rkf45" in l], flush=True); continue
        print(json.dumps({"system": name, "variant": tag, "calls_per_s_8_per_launch": rate(s, spec, 65536, spec.dt)}), flush=True)
