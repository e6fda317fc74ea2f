#!/usr/bin/env python3
"""stepHam calls/s of the one-trajectory-per-lane adaptive kernel with its stage vectors left to the register allocator
and parked in a run-time-indexed private array (hamk_options::rkf_park), one MI355X:
  python scripts/lane_rkf_park_ab.py [systems [B [which: 01 | 1 | 0]]] > gpurun_out/r03_lane_rkf_park.jsonl
(HAMK_HIPRTC_FLAGS=-DHAMK_RKF_LDS_BUDGET=34 in the environment: the parked stepper with LDS for two blocks per CU)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from hamilton_amd import _abi, api, examples

sys.path.insert(0, os.path.join(ROOT, "scripts"))
from quad_rkf_park_ab import stepham_rate                   # noqa: E402

NAMES = sys.argv[1].split(",") if len(sys.argv) > 1 else ["chain4", "chain5", "chain6", "threeBodyPolar", "chain7", "chain8", "chain9", "chain10", "chain12", "chain14", "chain16"]
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
PARKS = [int(c) for c in sys.argv[3]] if len(sys.argv) > 3 else [0, 1]
FLAGS = os.environ.get("HAMK_HIPRTC_FLAGS", "")
for name in NAMES:
    spec = examples.get(name)
    res = {}
    for park in PARKS:
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE, "rkf_park": _abi.ON if park else _abi.OFF})
        rate, nsub, out = stepham_rate(s, spec, BATCH, 4 * spec.dt)
        res[park] = out
        print(json.dumps({"what": "stepham", "system": name, "B": BATCH, "flags": FLAGS, "rkf_park": park, "calls_per_s": rate, "mean_substeps": nsub}), flush=True)
    if len(res) < 2:
        continue
    d = max(float((res[0].positions - res[1].positions).abs().max()), float((res[0].momenta - res[1].momenta).abs().max()))
    print(json.dumps({"what": "agreement", "system": name, "max_abs_diff_parked_vs_registers": d}), flush=True)
