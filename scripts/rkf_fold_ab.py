#!/usr/bin/env python3
"""Round 6, the parked adaptive stepper (hamk_device.hpp rkf45_body_parked), two changes each against the round-5 kernel on one box:
  fold  (n = 13..16)  stage 6's combinations folded in stage 5 (HAMK_RKF_FOLD): 13 scratch rows per attempt instead of 15
  frcp  (every n)     the error norm's 2n IEEE divisions as frcp + multiplication (HAMK_RKF_FRCP = 1: behind a range branch, 2: branch-free)
stepHam(dt) and stepHam(4 dt) at B = 65 536 from moving chains; sub-step counts of every variant must equal the round-5 kernel's.
  python scripts/rkf_fold_ab.py [--compile-only] > gpurun_out/r06_rkf_fold_ab.jsonl"""
import json
import os
os.environ["HAMK_TEST_OVERRIDES"] = "1"               # this script drives libhamk.so through its HAMK_* test overrides
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
COMPILE_ONLY = "--compile-only" in sys.argv
import numpy as np
from hamilton_amd import _abi, api, examples

OLD = "-DHAMK_RKF_FOLD=0 -DHAMK_RKF_FRCP=0"
PLAN = {}
for n in (13, 14, 15, 16):
    PLAN[f"chain{n}"] = [("round5", OLD), ("fold", "-DHAMK_RKF_FOLD=1 -DHAMK_RKF_FRCP=0"), ("frcp-branch", "-DHAMK_RKF_FOLD=0 -DHAMK_RKF_FRCP=1"),
                         ("frcp-select", "-DHAMK_RKF_FOLD=0 -DHAMK_RKF_FRCP=2"), ("fold+frcp-select", "-DHAMK_RKF_FOLD=1 -DHAMK_RKF_FRCP=2")]
for n in (8, 10, 12):
    PLAN[f"chain{n}"] = [("round5", OLD), ("frcp-branch", "-DHAMK_RKF_FRCP=1"), ("frcp-select", "-DHAMK_RKF_FRCP=2")]
PLAN["threeBodyPolar"] = [("round5", OLD), ("frcp-branch", "-DHAMK_RKF_FRCP=1"), ("frcp-select", "-DHAMK_RKF_FRCP=2")]
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
    return B / (best * 1e-3), s.last_nsub.clone(), out


for name, variants in PLAN.items():
    spec = examples.get(name)
    B = 65536 if spec.n >= 8 else 262144
    ref = {}
    for tag, flags in variants:
        os.environ["HAMK_HIPRTC_FLAGS"] = flags
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE})
        if COMPILE_ONLY:
            print(name, tag, [l for l in s.build_info.splitlines() if l.startswith("hamk_rkf45_k")], flush=True)
            continue
        for mult in (1, 4):
            rate, nsub, out = stepham_rate(s, spec, B, mult * spec.dt)
            rec = {"what": "stepham", "system": name, "B": B, "variant": tag, "flags": flags, "dt_mult": mult, "calls_per_s": rate, "mean_substeps": float(nsub.double().mean())}
            if mult not in ref:
                ref[mult] = (nsub, out)
            else:
                rec["identical_substep_counts_frac"] = float((nsub == ref[mult][0]).double().mean())
                rec["max_abs_diff_to_round5"] = float(max((out.positions - ref[mult][1].positions).abs().max(), (out.momenta - ref[mult][1].momenta).abs().max()))
            print(json.dumps(rec), flush=True)
