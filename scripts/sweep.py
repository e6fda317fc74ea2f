"""GPU: throughput of hamk_rk4_steps per system x code-generation variant
(HAMK_AD_MODE = H|D, HAMK_RK4_LOOP = 0|1; read by hamk_system_create)."""
import os, sys, time, json
os.environ["HAMK_TEST_OVERRIDES"] = "1"               # this script drives libhamk.so through its HAMK_* test overrides
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamilton_amd import api, examples as E

def run(name, B, nsteps, mode, loop, reps=3):
    os.environ["HAMK_AD_MODE"] = mode
    if loop: os.environ["HAMK_RK4_LOOP"] = loop
    else: os.environ.pop("HAMK_RK4_LOOP", None)
    spec = E.get(name)
    t0 = time.time(); s = api.system_from_spec(spec); tc = time.time() - t0
    q, qd = E.sample_config(spec, 0, B)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    st = api.Phase(ph.positions.clone(), ph.momenta.clone())
    api.rk4Steps(spec.dt, nsteps, s, st, inplace=True); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): api.rk4Steps(spec.dt, nsteps, s, st, inplace=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return dict(system=name, n=spec.n, m=spec.m, B=B, nsteps=nsteps, mode=mode, loop=loop, ms=ms,
                steps_per_s=B * nsteps / (ms * 1e-3), hbm_frac=B * nsteps / (ms * 1e-3) * 32 * spec.n / 8e12,
                compile_s=tc, flagged=int(torch.count_nonzero(s.last_status)))

if __name__ == "__main__":
    cfgs = [("doublePendulum", 1 << 20, 100), ("twoBody", 1 << 20, 100), ("spring", 1 << 20, 100),
            ("pendulum", 1 << 20, 100), ("room", 1 << 20, 100), ("bezier", 1 << 20, 100),
            ("threeBodyPolar", 1 << 18, 50), ("chain4", 1 << 16, 50), ("chain8", 1 << 16, 20), ("chain12", 1 << 16, 20),
            ("chain16", 1 << 16, 10), ("opcodeZoo", 1 << 18, 20)]
    only = sys.argv[1:]
    for name, B, ns in cfgs:
        if only and name not in only: continue
        for mode in ("H", "D", "R"):
            if mode == "H" and E.get(name).n > 6: continue
            for loop in (("0", "1") if os.environ.get("SWEEP_LOOPS") else ("",)):
                try:
                    print(json.dumps(run(name, B, ns, mode, loop)), flush=True)
                except Exception as ex:
                    print(json.dumps(dict(system=name, mode=mode, loop=loop, error=str(ex)[:300])), flush=True)
