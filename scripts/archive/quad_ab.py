#!/usr/bin/env python3
"""A/B measurements for the four-lane kernels on one MI355X (scripts/gpu_r03_d.sh):
  * RK4 steps/s of chain32 / chain24 / chain16 with the left-looking and the right-looking factorisation (HAMK_QUAD_LEFT),
  * stepHam calls/s of chain20 / chain32 on the quad and on the wave-cooperative module,
  * stepHam calls/s of chain48 with the n > 32 adaptive kernel at one and at two wavefronts per SIMD."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
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


def stepham_rate(s, spec, B, dt):
    q, qd = examples.sample_config(spec, 0, B)
    qd = 0.3 * np.cos(np.arange(spec.n * B).reshape(spec.n, B) * 0.7)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    st = api.Phase(ph.positions.clone(), ph.momenta.clone())
    api.stepHam(dt, s, st)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); api.stepHam(dt, s, st); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return B / (best * 1e-3), float(s.last_nsub.double().mean())


out = []
for name, B, nsteps in (("chain32", 65536, 100), ("chain32", 16384, 200), ("chain24", 65536, 100), ("chain16", 16384, 400), ("chain16", 65536, 200)):
    spec = examples.get(name)
    for left in (1, 0):
        os.environ["HAMK_HIPRTC_FLAGS"] = f"-DHAMK_QUAD_LEFT={left}"
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_QUAD})
        r = {"what": "rk4", "system": name, "B": B, "quad_left": left, "steps_per_s": rk4_rate(s, spec, B, nsteps)}
        print(json.dumps(r), flush=True); out.append(r)
os.environ.pop("HAMK_HIPRTC_FLAGS", None)
for name, B in (("chain20", 16384), ("chain32", 16384)):
    spec = examples.get(name)
    for label, mp in (("quad", _abi.MAP_QUAD), ("wave", _abi.MAP_WAVE)):
        s = api.system_from_spec(spec, {"mapping": mp})
        rate, nsub = stepham_rate(s, spec, B, 4 * spec.dt)
        r = {"what": "stepham", "system": name, "B": B, "mapping": label, "calls_per_s": rate, "mean_substeps": nsub}
        print(json.dumps(r), flush=True); out.append(r)
spec = examples.get("chain48")
for waves in (1, 2):
    s = api.system_from_spec(spec, {"mapping": _abi.MAP_WAVE, "rk4_min_waves": waves})
    rate, nsub = stepham_rate(s, spec, 8192, 4 * spec.dt)
    r = {"what": "stepham", "system": "chain48", "B": 8192, "waves_per_simd_cap": waves, "calls_per_s": rate, "mean_substeps": nsub}
    print(json.dumps(r), flush=True); out.append(r)
