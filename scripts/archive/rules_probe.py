#!/usr/bin/env python3
"""Data behind two dispatch rules of hamk_api.cpp on one MI355X:
  * rk4_park (RK4 state parked in LDS) (AUTO: 14 <= n <= 16): chain10/12/14/16 with the parking on and off at B = 65 536;
  * quad_below (four lanes per trajectory for small ensembles, n >= 12): chain10/12/14 on the lane and quad kernels at
    B = 8 192 / 16 384 / 32 768.
  python scripts/rules_probe.py > gpurun_out/r03_rules_probe.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hamilton_amd import _abi, api, examples


def rate(s, spec, B, nsteps):
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


for name in ("chain10", "chain12", "chain14", "chain16"):
    spec = examples.get(name)
    for park in (_abi.ON, _abi.OFF):
        try:
            s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE, "rk4_park": park})
            r = {"what": "park", "system": name, "park": park == _abi.ON, "B": 65536, "steps_per_s": rate(s, spec, 65536, 200), "trig": s.options()["trig"]}
        except Exception as e:                               # noqa: BLE001
            r = {"what": "park", "system": name, "park": park == _abi.ON, "error": repr(e)[:200]}
        print(json.dumps(r), flush=True)
for name in ("chain10", "chain12", "chain14"):
    spec = examples.get(name)
    for label, mp in (("lane", _abi.MAP_LANE), ("quad", _abi.MAP_QUAD)):
        s = api.system_from_spec(spec, {"mapping": mp})
        for B in (8192, 16384, 32768):
            print(json.dumps({"what": "mapping", "system": name, "mapping": label, "B": B, "steps_per_s": rate(s, spec, B, 400)}), flush=True)
