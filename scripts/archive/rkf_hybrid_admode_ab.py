import json, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts/archive")
COMPILE_ONLY = "--compile-only" in sys.argv
import numpy as np
from hamilton_amd import _abi, api, examples
if not COMPILE_ONLY:
    import torch
def rate(s, spec, B, dt):
    q, qd = examples.sample_config(spec, 0, B)
    if spec.name.startswith("chain"):
        qd = 0.3 * np.cos(np.arange(spec.n * B).reshape(spec.n, B) * 0.7)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    st = api.Phase(ph.positions.clone(), ph.momenta.clone())
    api.stepHam(dt, s, st); torch.cuda.synchronize()
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); api.stepHam(dt, s, st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1); best = ms if best is None else min(best, ms)
    return B / (best * 1e-3)
for name in ("threeBodyPolar", "chain6", "chain7", "chain5"):
    spec = examples.get(name)
    for mode in (None, "R", "D"):
        if mode: os.environ["HAMK_AD_MODE"] = mode
        else: os.environ.pop("HAMK_AD_MODE", None)
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE})
        if COMPILE_ONLY:
            print(name, mode, [l for l in s.build_info.splitlines() if "rkf45" in l]); continue
        print(json.dumps({"system": name, "ad_mode": mode or "auto", "calls_per_s": rate(s, spec, 262144, spec.dt)}), flush=True)
