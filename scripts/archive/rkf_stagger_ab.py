#!/usr/bin/env python3
"""Parked adaptive stepper (lane kernels): blocks started out of phase (-DHAMK_RKF_STAGGER=k: (blockIdx & 3) x k x s_sleep 127)
against all in step, one MI355X.  python scripts/rkf_stagger_ab.py [--compile-only] > gpurun_out/r04_rkf_stagger_ab.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
COMPILE_ONLY = "--compile-only" in sys.argv
import numpy as np
from hamilton_amd import _abi, api, examples

PLAN = {"chain16": (0, 1, 2, 4), "chain14": (0, 1, 2), "chain12": (0, 1, 2), "chain10": (0, 1), "chain8": (0, 1), "threeBodyPolar": (0, 1)}
if not COMPILE_ONLY:
    import torch


def stepham_rate(s, spec, B, dt):
    q, qd = examples.sample_config(spec, 0, B)
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


for name, ks in PLAN.items():
    spec = examples.get(name)
    B = 262144 if name == "threeBodyPolar" else 65536
    ref = None
    for k in ks:
        os.environ["HAMK_HIPRTC_FLAGS"] = f"-DHAMK_RKF_STAGGER={k}"
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE})
        if COMPILE_ONLY:
            print(name, k, [l for l in s.build_info.splitlines() if l.startswith("hamk_rkf45_k")], flush=True)
            continue
        for mult in (1, 4):
            rate, nsub, out = stepham_rate(s, spec, B, mult * spec.dt)
            rec = {"what": "stepham", "system": name, "B": B, "stagger": k, "dt_mult": mult, "calls_per_s": rate, "mean_substeps": nsub}
            if mult == 1:
                if ref is None:
                    ref = out
                else:
                    rec["bit_identical_to_stagger_0"] = bool(torch.equal(out.positions, ref.positions) and torch.equal(out.momenta, ref.momenta))
            print(json.dumps(rec), flush=True)
