#!/usr/bin/env python3
"""The sizes between the measured points of scripts/rules_probe.py: chain13 with the RK4 state parked in LDS on / off at
B = 65 536 (rule: from n = 14), chain11 on the lane and quad kernels at B = 8 192 ... 32 768 (rule: quad below 32 768 from
n = 12), chain13 the same.
  python scripts/rules_probe2.py >> gpurun_out/r03_rules_probe.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hamilton_amd import _abi, api, examples


def rk4_rate(s, spec, B, nsteps):
    q, qd = examples.sample_config(spec, 0, B)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    st = api.Phase(ph.positions.clone(), ph.momenta.clone())
    api.rk4Steps(spec.dt, 4, s, st, inplace=True)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); api.rk4Steps(spec.dt, nsteps, s, st, inplace=True); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return B * nsteps / (best * 1e-3)


spec = examples.get("chain13")
for park in (_abi.ON, _abi.OFF):
    s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE, "rk4_park": park})
    print(json.dumps({"what": "park", "system": "chain13", "park": park == _abi.ON, "B": 65536, "steps_per_s": rk4_rate(s, spec, 65536, 200)}), flush=True)
for name in ("chain11", "chain13"):
    spec = examples.get(name)
    for label, mp in (("lane", _abi.MAP_LANE), ("quad", _abi.MAP_QUAD)):
        s = api.system_from_spec(spec, {"mapping": mp})
        for B in (8192, 16384, 32768):
            print(json.dumps({"what": "mapping", "system": name, "mapping": label, "B": B, "steps_per_s": rk4_rate(s, spec, B, 400)}), flush=True)
