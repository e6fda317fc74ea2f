#!/usr/bin/env python3
"""Which mapping the ADAPTIVE kernel of a mid-size system should use for a small ensemble: stepHam calls/s of chain12/14/16
at B = 8 192 / 16 384 / 32 768 on the lane kernels (parked stepper) and on the quad kernels with and without the parked
stepper -- the data behind hamk_api.cpp choose_mapping for K_RKF45.
  python scripts/stepham_small_b.py > gpurun_out/r03_stepham_small_b.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from hamilton_amd import _abi, api, examples
from quad_rkf_park_ab import stepham_rate                   # noqa: E402

for name in ("chain12", "chain14", "chain16"):
    spec = examples.get(name)
    for label, opt in (("lane parked", {"mapping": _abi.MAP_LANE}), ("quad", {"mapping": _abi.MAP_QUAD, "rkf_park": _abi.OFF}),
                       ("quad parked", {"mapping": _abi.MAP_QUAD, "rkf_park": _abi.ON})):
        s = api.system_from_spec(spec, opt)
        for B in (8192, 16384, 32768):
            rate, nsub, _ = stepham_rate(s, spec, B, 4 * spec.dt)
            print(json.dumps({"what": "stepham", "system": name, "kernels": label, "B": B, "calls_per_s": rate, "mean_substeps": nsub}), flush=True)
