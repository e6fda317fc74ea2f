import json, os, sys
os.environ["HAMK_TEST_OVERRIDES"]="1"; os.environ["HAMK_SELFCHECK"]="0"
sys.path.insert(0, os.getcwd())
os.environ.setdefault("HAMK_CACHE_DIR", os.path.join(os.getcwd(), ".hamk_cache"))
from hamilton_amd import _abi, api, examples
COMPILE_ONLY = "--compile-only" in sys.argv
if not COMPILE_ONLY:
    import torch
for name in ("chain40", "chain48", "chain64", "dense32"):
    spec = examples.get(name)
    B = 16384
    ref = None
    for waves in (2, 1):
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_WAVE, "rk4_min_waves": waves})
        info = [l for l in s.build_info.splitlines() if l.startswith("hamk_rkf45_k")]
        if COMPILE_ONLY:
            print(name, waves, info, flush=True); continue
        q, qd = examples.sample_config(spec, 0, B)
        ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
        dt = spec.dt
        out = api.stepHam(dt, s, ph); torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = api.stepHam(dt, s, ph); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1); best = ms if best is None else min(best, ms)
        rec = {"system": name, "rk4_min_waves": waves, "B": B, "ms": best, "calls_per_s": B / (best * 1e-3), "mean_substeps": float(s.last_nsub.double().mean()), "build": info[0] if info else None}
        if ref is None: ref = out
        else: rec["max_abs_diff_to_w2"] = float(max((out.positions - ref.positions).abs().max(), (out.momenta - ref.momenta).abs().max()))
        print(json.dumps(rec), flush=True)
