#!/usr/bin/env python3
"""Round 6: the table evaluations of a right-hand side in ONE burst (hamk_device.hpp trig_burst_lut: all reductions, all gathers, the
kernels in r, then the angle additions) against one site at a time (-DHAMK_TRIG_BURST=0, the round's earlier kernels): lane kernels of
the chains, RK4 steps/s and stepHam calls/s, same box, same state; results must agree bit for bit.
  python scripts/trig_burst_ab.py [--compile-only]"""
import json
import os
os.environ["HAMK_TEST_OVERRIDES"] = "1"
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
COMPILE_ONLY = "--compile-only" in sys.argv
from hamilton_amd import _abi, api, examples
if not COMPILE_ONLY:
    import numpy as np
    import torch

SYSTEMS = (("chain5", 1 << 17, 400), ("chain6", 1 << 17, 400), ("chain7", 1 << 16, 400), ("chain8", 1 << 16, 400), ("chain9", 1 << 16, 300), ("chain10", 1 << 16, 300), ("chain11", 1 << 16, 200),
           ("chain12", 1 << 16, 200), ("chain13", 1 << 16, 200), ("chain14", 1 << 16, 100), ("chain15", 1 << 16, 100), ("chain16", 1 << 16, 100))
VARIANTS = (("one site at a time", "-DHAMK_TRIG_BURST=0"), ("burst", ""))


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for name, B, nsteps in SYSTEMS:
    spec = examples.get(name)
    built = []
    for tag, flags in VARIANTS:
        os.environ["HAMK_HIPRTC_FLAGS"] = flags
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE})
        lines = [l for l in s.build_info.splitlines() if l.startswith(("hamk_rk4_steps_k", "hamk_rkf45_k"))]
        if COMPILE_ONLY:
            print(name, tag, lines, flush=True)
        built.append((tag, flags, s, lines))
    if COMPILE_ONLY:
        continue
    q, qd = examples.sample_config(spec, 0, B)
    ph = api.toPhase(built[0][2], api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    outs = []
    for tag, flags, s, lines in built:
        out = api.rk4Steps(spec.dt, 20, s, api.Phase(ph.positions.clone(), ph.momenta.clone()))
        sh = api.stepHam(spec.dt, s, api.Phase(ph.positions.clone(), ph.momenta.clone()))
        outs.append((out, sh))
    diff = float(max((outs[0][0].positions - outs[1][0].positions).abs().max(), (outs[0][0].momenta - outs[1][0].momenta).abs().max(),
                     (outs[0][1].positions - outs[1][1].positions).abs().max(), (outs[0][1].momenta - outs[1][1].momenta).abs().max()))
    # the variants take turns (three rounds of five launches each): clock and box drift hit both alike
    best_rk4, best_sh = [None, None], [None, None]
    states = [api.Phase(ph.positions.clone(), ph.momenta.clone()) for _ in built]
    for rnd in range(3):
        for i, (tag, flags, s, lines) in enumerate(built):
            for _ in range(5):
                ms = timed(lambda: api.rk4Steps(spec.dt, nsteps, s, states[i], inplace=True))
                best_rk4[i] = ms if best_rk4[i] is None else min(best_rk4[i], ms)
            for _ in range(5):
                st2 = api.Phase(ph.positions.clone(), ph.momenta.clone())
                ms = timed(lambda: api.iterateStepHam(spec.dt, 8, s, st2, inplace=True))
                best_sh[i] = ms if best_sh[i] is None else min(best_sh[i], ms)
    for i, (tag, flags, s, lines) in enumerate(built):
        print(json.dumps({"what": "trig_burst_ab", "system": name, "B": B, "variant": tag, "flags": flags, "rk4_steps_per_s": B * nsteps / (best_rk4[i] * 1e-3),
                          "stepham_calls_per_s": 8 * B / (best_sh[i] * 1e-3), "max_abs_diff_between_variants_20_rk4_steps_and_one_stepham": diff, "build": lines}), flush=True)
