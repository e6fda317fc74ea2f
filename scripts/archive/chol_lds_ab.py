#!/usr/bin/env python3
"""Round 6, review item 2 (chain32 on the four-lane kernels: a BUILT alternative for the factorisation's broadcasts).
The left-looking panel update of hamk_quad.hpp chol needs, per finished column, the panel's four row entries in all four lanes: 8 DPP
moves per column in the shipped kernel.  -DHAMK_CHOL_LDS=k takes them through LDS from panel k on (half a ds_write_b128 + two
ds_read_b128 per column; chunks written a panel ahead, loads HAMK_CHOL_LDS_PD column pairs ahead): 6 377 -> ~5 800 VALU instructions per
stage at n = 32, +210 LDS instructions.  RK4 steps/s of each variant, same box, same state; results against the default build.
  python scripts/archive/chol_lds_ab.py [--compile-only]      (needs the tree of commit bdf3ae2: the variant was deleted after the measurement)"""
import json
import os
os.environ["HAMK_TEST_OVERRIDES"] = "1"
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
COMPILE_ONLY = "--compile-only" in sys.argv
from hamilton_amd import _abi, api, examples
if not COMPILE_ONLY:
    import numpy as np
    import torch

VARIANTS = (("dpp (shipped)", "", None),
            ("lds from panel 2, 2 pairs ahead", "-DHAMK_CHOL_LDS=1", "0"),
            ("lds from panel 2, 1 pair ahead", "-DHAMK_CHOL_LDS=1 -DHAMK_CHOL_LDS_PD=1", "0"),
            ("lds panels 2..6", "-DHAMK_CHOL_LDS=1 -DHAMK_CHOL_LDS_LAST=6", "0"),
            ("lds panels 2..5", "-DHAMK_CHOL_LDS=1 -DHAMK_CHOL_LDS_LAST=5", "0"),
            ("lds panels 4..7", "-DHAMK_CHOL_LDS=4", "0"))
SYSTEMS = (("chain32", 1 << 16, 100), ("chain24", 1 << 16, 100), ("chain20", 1 << 16, 100))
if "--chain32-only" in sys.argv:
    SYSTEMS = SYSTEMS[:1]

for name, B, nsteps in SYSTEMS:
    spec = examples.get(name)
    ref = None
    for tag, flags, park in VARIANTS:
        if name != "chain32" and "panels" in tag:
            continue
        os.environ["HAMK_HIPRTC_FLAGS"] = flags
        if park is not None and name == "chain32":
            os.environ["HAMK_RKF_PARK"] = park            # (n = 32: the parked stepper's rows do not fit next to the exchange region; only the RK4 kernel is measured)
        else:
            os.environ.pop("HAMK_RKF_PARK", None)
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_QUAD})
        line = [l for l in s.build_info.splitlines() if l.startswith("hamk_rk4_steps_k")]
        if COMPILE_ONLY:
            print(name, tag, line, flush=True)
            continue
        q, qd = examples.sample_config(spec, 0, B)
        ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
        st = api.Phase(ph.positions.clone(), ph.momenta.clone())
        out = api.rk4Steps(spec.dt, 20, s, st)
        if ref is None:
            ref = out
        api.rk4Steps(spec.dt, nsteps, s, st, inplace=True)
        torch.cuda.synchronize()
        best = None
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); api.rk4Steps(spec.dt, nsteps, s, st, inplace=True); e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        print(json.dumps({"what": "chol_lds_ab", "system": name, "B": B, "variant": tag, "flags": flags, "rk4_steps_per_s": B * nsteps / (best * 1e-3),
                          "ms_per_launch": best, "build": line[0] if line else None,
                          "max_abs_diff_after_20_steps_vs_shipped": float(max((out.positions - ref.positions).abs().max(), (out.momenta - ref.momenta).abs().max()))}), flush=True)
