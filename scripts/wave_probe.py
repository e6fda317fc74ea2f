#!/usr/bin/env python3
"""GPU: the wave-cooperative RK4 kernel on this round's tree -- dense-Jacobian systems and chains beyond the
four-lane mapping: RK4 steps/s at one and two wavefronts per SIMD (hamk_options::rk4_min_waves), and where the time of a
right-hand side goes (one phase removed at a time: -DHAMK_PROBE_SKIP_* in hamk_wave.hpp; those builds are timing only).
  python scripts/wave_probe.py [--compile-only] [--systems=dense32,chain64] [--attrib] > gpurun_out/r05_wave_probe.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HAMK_TEST_OVERRIDES"] = "1"
os.environ.setdefault("HAMK_CACHE_DIR", os.path.join(ROOT, ".hamk_cache"))
os.environ["HAMK_SELFCHECK"] = "0"
COMPILE_ONLY = "--compile-only" in sys.argv
from hamilton_amd import _abi, api, examples                # noqa: E402

SYSTEMS = ["dense24", "dense32", "chain32", "chain48", "chain64"]
for a in sys.argv[1:]:
    if a.startswith("--systems="):
        SYSTEMS = a.split("=", 1)[1].split(",")
ATTRIB = [("-kacc", "-DHAMK_PROBE_SKIP_KACC"), ("-factor", "-DHAMK_PROBE_SKIP_FACTOR"), ("-solve", "-DHAMK_PROBE_SKIP_SOLVE"),
          ("-sweep2", "-DHAMK_PROBE_SKIP_SWEEP2"),
          ("-all4", "-DHAMK_PROBE_SKIP_KACC -DHAMK_PROBE_SKIP_FACTOR -DHAMK_PROBE_SKIP_SOLVE -DHAMK_PROBE_SKIP_SWEEP2")] if "--attrib" in sys.argv else []
EXTRA = []
for a in sys.argv[1:]:
    if a.startswith("--variant="):                          # --variant=tag:flags ("_-D" for " -D")
        tag, fl = a.split("=", 1)[1].split(":", 1)
        EXTRA.append((tag, fl.replace("_-D", " -D")))
if not COMPILE_ONLY:
    import torch


def build(spec, flags, waves):
    os.environ["HAMK_HIPRTC_FLAGS"] = flags
    return api.system_from_spec(spec, {"mapping": _abi.MAP_WAVE, "rk4_min_waves": waves})


def rate(s, spec, B, nsteps):
    q, qd = examples.sample_config(spec, 0, B)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    st = api.Phase(ph.positions.clone(), ph.momenta.clone())
    api.rk4Steps(spec.dt, nsteps, s, st, inplace=True)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); api.rk4Steps(spec.dt, nsteps, s, st, inplace=True); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best, st


for name in SYSTEMS:
    spec = examples.get(name)
    B, nsteps = 16384, 10
    plan = [("w2", "", 2), ("w1", "", 1)] + [(t, f, 2) for t, f in ATTRIB + EXTRA]
    ref = None
    for tag, flags, waves in plan:
        s = build(spec, flags, waves)
        info = [l for l in s.build_info.splitlines() if l.startswith("hamk_rk4_steps_k")]
        if COMPILE_ONLY:
            print(name, tag, info, flush=True)
            continue
        ms, st = rate(s, spec, B, nsteps)
        rec = {"system": name, "variant": tag, "flags": flags, "rk4_min_waves": waves, "B": B, "nsteps": nsteps, "ms": ms, "steps_per_s": B * nsteps / (ms * 1e-3),
               "build": info[0] if info else None}
        if not flags:
            if ref is None:
                ref = st
            else:
                rec["max_abs_diff_to_w2"] = float(max((st.positions - ref.positions).abs().max(), (st.momenta - ref.momenta).abs().max()))
        print(json.dumps(rec), flush=True)
