#!/usr/bin/env python3
"""Parked adaptive stepper for n = 6, 7 at TWO wavefronts per SIMD: y / dydt (+ one row) in LDS, the other rows in REGISTERS
(-DHAMK_RKF_ROWS_IN_REGS=1: static indices, the private array is promoted), against the shipped one-wavefront version with
every row in LDS.  python scripts/rkf_hybrid_ab.py [--compile-only] > gpurun_out/r04_rkf_hybrid_ab.jsonl"""
import json
import os
os.environ["HAMK_TEST_OVERRIDES"] = "1"               # this script drives libhamk.so through its HAMK_* test overrides
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
COMPILE_ONLY = "--compile-only" in sys.argv
import numpy as np
from hamilton_amd import _abi, api, examples

R = "-DHAMK_RKF_ROWS_IN_REGS=1 -DHAMK_RKF_MIN_WAVES_LANE=2"
PLAN = {
    "threeBodyPolar": [("shipped", "", None), ("regs-w2-b36", R + " -DHAMK_RKF_LDS_BUDGET=36", None), ("regs-w2-b36-notable", R + " -DHAMK_RKF_LDS_BUDGET=36", "0"),
                       ("regs-w2-b24", R + " -DHAMK_RKF_LDS_BUDGET=24", None), ("regs-w2-b24-notable", R + " -DHAMK_RKF_LDS_BUDGET=24", "0"),
                       ("regs-w3-b24-notable", "-DHAMK_RKF_ROWS_IN_REGS=1 -DHAMK_RKF_MIN_WAVES_LANE=3 -DHAMK_RKF_LDS_BUDGET=24", "0")],
    "chain6": [("shipped", "", None), ("regs-w2-b36", R + " -DHAMK_RKF_LDS_BUDGET=36", None), ("regs-w2-b36-notable", R + " -DHAMK_RKF_LDS_BUDGET=36", "0")],
    "chain7": [("shipped", "", None), ("regs-w2-b28", R + " -DHAMK_RKF_LDS_BUDGET=28", None), ("regs-w2-b28-notable", R + " -DHAMK_RKF_LDS_BUDGET=28", "0")],
    "chain5": [("shipped", "", None), ("regs-w2-b30", R + " -DHAMK_RKF_LDS_BUDGET=30", None)],
    "chain4": [("shipped", "", None), ("regs-w2-b32", R + " -DHAMK_RKF_LDS_BUDGET=32", None)],
}
if not COMPILE_ONLY:
    import torch


def stepham_rate(s, spec, B, dt):
    q, qd = examples.sample_config(spec, 0, B)
    if spec.name.startswith("chain"):
        qd = 0.3 * np.cos(np.arange(spec.n * B).reshape(spec.n, B) * 0.7)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    st = api.Phase(ph.positions.clone(), ph.momenta.clone())
    out = api.stepHam(dt, s, st)
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = api.stepHam(dt, s, st); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return B / (best * 1e-3), float(s.last_nsub.double().mean()), out


for name, variants in PLAN.items():
    spec = examples.get(name)
    B = 262144
    ref = None
    for tag, flags, lut in variants:
        os.environ["HAMK_HIPRTC_FLAGS"] = flags
        if lut is None:
            os.environ.pop("HAMK_TRIG_LUT", None)
        else:
            os.environ["HAMK_TRIG_LUT"] = lut
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE})
        if COMPILE_ONLY:
            print(name, tag, [l for l in s.build_info.splitlines() if l.startswith("hamk_rkf45_k")], flush=True)
            continue
        for mult in (1, 4):
            rate, nsub, out = stepham_rate(s, spec, B, mult * spec.dt)
            rec = {"what": "stepham", "system": name, "B": B, "variant": tag, "flags": flags, "trig_lut": lut, "dt_mult": mult, "calls_per_s": rate, "mean_substeps": nsub}
            if mult == 1:
                if ref is None:
                    ref = out
                else:
                    rec["max_abs_diff_to_shipped"] = float(max((out.positions - ref.positions).abs().max(), (out.momenta - ref.momenta).abs().max()))
            print(json.dumps(rec), flush=True)
