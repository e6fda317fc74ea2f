#!/usr/bin/env python3
"""Parked adaptive stepper, mid-size systems at ONE wavefront per SIMD: the rows beyond the LDS share in registers
(-DHAMK_RKF_ROWS_IN_REGS=1) instead of scratch.  python scripts/archive/rkf_rows_in_regs_ab.py [--compile-only]"""
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
    out = api.stepHam(dt, s, st); torch.cuda.synchronize()
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = api.stepHam(dt, s, st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1); best = ms if best is None else min(best, ms)
    return B / (best * 1e-3), out
for name in ("chain8", "chain9", "chain10", "chain11", "chain12", "chain13"):
    spec = examples.get(name)
    ref = None
    for tag, flags in (("scratch", ""), ("regs", "-DHAMK_RKF_ROWS_IN_REGS=1")):
        os.environ["HAMK_HIPRTC_FLAGS"] = flags
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE})
        if COMPILE_ONLY:
            print(name, tag, [l for l in s.build_info.splitlines() if "rkf45" in l], flush=True); continue
        for mult in (1, 4):
            r, out = rate(s, spec, 65536, mult * spec.dt)
            rec = {"system": name, "variant": tag, "dt_mult": mult, "calls_per_s": r}
            if mult == 1:
                if ref is None: ref = out
                else: rec["max_abs_diff"] = float(max((out.positions - ref.positions).abs().max(), (out.momenta - ref.momenta).abs().max()))
            print(json.dumps(rec), flush=True)
