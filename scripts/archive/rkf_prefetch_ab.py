#!/usr/bin/env python3
"""The parked adaptive stepper of the lane kernels (hamk_device.hpp rkf45_body_parked), variants A/B on one MI355X:
  rows R   -DHAMK_RKF_PREFETCH_ROWS=R: rows of the next stage combination fetched from scratch INSIDE the right-hand side
           (0 = round 3: loaded at the top of the stage, 1..3)
  w2/bL    two wavefronts per SIMD (-DHAMK_RKF_MIN_WAVES_LANE=2) with an LDS budget of L doubles per lane
  python scripts/rkf_prefetch_ab.py [--compile-only] > gpurun_out/r04_rkf_prefetch_ab.jsonl
Every variant must give the SAME bits (only when loads are issued and where rows wait changes): checked against the first."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
COMPILE_ONLY = "--compile-only" in sys.argv
import numpy as np
from hamilton_amd import _abi, api, examples

ROWS = {r: f"-DHAMK_RKF_PREFETCH_ROWS={r}" for r in (0, 1, 2, 3)}
PLAN = {
    "chain16": [("rows0", ROWS[0]), ("rows1", ROWS[1]), ("rows2", ROWS[2]), ("rows3", ROWS[3])],
    "chain14": [("rows0", ROWS[0]), ("rows2", ROWS[2]), ("rows3", ROWS[3])],
    "chain12": [("rows0", ROWS[0]), ("rows2", ROWS[2]), ("rows3", ROWS[3])],
    "chain10": [("rows0", ROWS[0]), ("rows3", ROWS[3])],
    "chain8": [("rows0", ROWS[0]), ("rows3", ROWS[3])],
    "threeBodyPolar": [("rows0", ROWS[0]), ("rows3", ROWS[3]), ("rows3-NL6", ROWS[3] + " -DHAMK_RKF_LDS_BUDGET=76"),
                       ("rows3-w2-b36", ROWS[3] + " -DHAMK_RKF_MIN_WAVES_LANE=2 -DHAMK_RKF_LDS_BUDGET=36"),
                       ("rows3-w2-b24", ROWS[3] + " -DHAMK_RKF_MIN_WAVES_LANE=2 -DHAMK_RKF_LDS_BUDGET=24"),
                       ("rows0-w2-b24", ROWS[0] + " -DHAMK_RKF_MIN_WAVES_LANE=2 -DHAMK_RKF_LDS_BUDGET=24")],
    "chain6": [("rows0", ROWS[0]), ("rows3", ROWS[3]), ("rows3-w2-b36", ROWS[3] + " -DHAMK_RKF_MIN_WAVES_LANE=2 -DHAMK_RKF_LDS_BUDGET=36")],
    "chain7": [("rows0", ROWS[0]), ("rows3", ROWS[3]), ("rows3-w2-b28", ROWS[3] + " -DHAMK_RKF_MIN_WAVES_LANE=2 -DHAMK_RKF_LDS_BUDGET=28")],
}
# (threeBodyPolar's default budget, 76 doubles, now gives NL = 6 = "rows3-NL6"; "rows0"/"rows3" there pin round 3's NL = 5)
for k in ("threeBodyPolar", "chain6"):
    PLAN[k] = [(n, f if "BUDGET" in f else f + " -DHAMK_RKF_LDS_BUDGET=60") for n, f in PLAN[k]]

if not COMPILE_ONLY:
    import torch
    from quad_rkf_park_ab import stepham_rate

names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(PLAN)
for name in names:
    spec = examples.get(name)
    B = 262144 if name == "threeBodyPolar" else 65536
    ref = None
    for tag, flags in PLAN[name]:
        os.environ["HAMK_HIPRTC_FLAGS"] = flags
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE, "rkf_park": _abi.ON})
        if COMPILE_ONLY:
            info = [l for l in s.build_info.splitlines() if l.startswith("hamk_rkf45_k")]
            print(name, tag, info, flush=True)
            continue
        for mult in (1, 4):
            rate, nsub, out = stepham_rate(s, spec, B, mult * spec.dt)
            rec = {"what": "stepham", "system": name, "B": B, "variant": tag, "flags": flags, "dt_mult": mult, "calls_per_s": rate, "mean_substeps": nsub}
            if mult == 1:
                if ref is None:
                    ref = out
                else:
                    rec["bit_identical_to_first_variant"] = bool(torch.equal(out.positions, ref.positions) and torch.equal(out.momenta, ref.momenta))
            print(json.dumps(rec), flush=True)
