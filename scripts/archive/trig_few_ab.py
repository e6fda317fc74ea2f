#!/usr/bin/env python3
"""Round 6: rotations (one table evaluation per step + three rotations per sincos site) for systems with MORE than four sites -- the chains
of 5 ... 8 links, whose stepping kernels run one wavefront per SIMD and wait for their table gathers (chain8: SQ_WAIT_ANY 0.20) --
against every evaluation through the LDS table.  -DHAMK_TRIG_FEW_MAX=8 with HAMK_TRIG_LUT=2 vs the default; RK4 steps/s, same box.
  python scripts/archive/trig_few_ab.py [--compile-only]"""
import json
import os
os.environ["HAMK_TEST_OVERRIDES"] = "1"
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
COMPILE_ONLY = "--compile-only" in sys.argv
from hamilton_amd import api, examples
if not COMPILE_ONLY:
    import numpy as np
    import torch

for name, B in (("chain8", 1 << 16), ("chain6", 1 << 17), ("chain5", 1 << 17)):
    spec = examples.get(name)
    ref = None
    for tag, flags, lut in (("table", "", None), ("rotations", "-DHAMK_TRIG_FEW_MAX=8", "2")):
        os.environ["HAMK_HIPRTC_FLAGS"] = flags
        if lut:
            os.environ["HAMK_TRIG_LUT"] = lut
        else:
            os.environ.pop("HAMK_TRIG_LUT", None)
        s = api.system_from_spec(spec)
        if COMPILE_ONLY:
            print(name, tag, [l for l in s.build_info.splitlines() if l.startswith("hamk_rk4_steps_k")], flush=True)
            continue
        q, qd = examples.sample_config(spec, 0, B)
        ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
        st = api.Phase(ph.positions.clone(), ph.momenta.clone())
        out = api.rk4Steps(spec.dt, 50, s, st)
        if ref is None:
            ref = out
        api.rk4Steps(spec.dt, 200, s, st, inplace=True)
        torch.cuda.synchronize()
        best = None
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); api.rk4Steps(spec.dt, 200, s, st, inplace=True); e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        print(json.dumps({"what": "trig_few_ab", "system": name, "B": B, "variant": tag, "rk4_steps_per_s": B * 200 / (best * 1e-3),
                          "max_abs_diff_after_50_steps_vs_table": float(max((out.positions - ref.positions).abs().max(), (out.momenta - ref.momenta).abs().max()))}), flush=True)
