"""GPU: throughput of the wave-cooperative path (BASELINE.json config 5: N-link chain,
B = 65,536, dt = 0.005) and the lane path on the sizes both support."""
import os, sys, time, json
os.environ["HAMK_TEST_OVERRIDES"] = "1"               # this script drives libhamk.so through its HAMK_* test overrides
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamilton_amd import api, examples as E

def run(name, B, nsteps, wave, reps=3):
    if wave is None: os.environ.pop("HAMK_WAVE", None)
    else: os.environ["HAMK_WAVE"] = wave
    spec = E.get(name)
    t0 = time.time(); s = api.system_from_spec(spec); tc = time.time() - t0
    q, qd = E.sample_config(spec, 0, B)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    st = api.Phase(ph.positions.clone(), ph.momenta.clone())
    h0 = api.hamiltonian(s, st).clone()
    api.rk4Steps(spec.dt, nsteps, s, st, inplace=True); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): api.rk4Steps(spec.dt, nsteps, s, st, inplace=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    drift = float(((api.hamiltonian(s, st) - h0).abs() / h0.abs().clamp(min=1.0)).max())
    return dict(system=name, n=spec.n, m=spec.m, B=B, nsteps=nsteps, path="wave" if "INSTANTIATE_WAVE" in s.source else "lane",
                ms=round(ms, 3), steps_per_s=B * nsteps / (ms * 1e-3), hbm_frac=round(B * nsteps / (ms * 1e-3) * 32 * spec.n / 8e12, 4),
                compile_s=round(tc, 1), flagged=int(torch.count_nonzero(s.last_status)), max_rel_energy_drift=drift)

if __name__ == "__main__":
    cfgs = [("chain8", 65536, 50, "0"), ("chain8", 65536, 50, "1"), ("chain12", 65536, 20, "0"), ("chain12", 65536, 20, "1"),
            ("chain16", 65536, 20, "0"), ("chain16", 65536, 20, "1"), ("chain20", 65536, 10, None), ("chain32", 65536, 10, None)]
    only = sys.argv[1:]
    for name, B, ns, wave in cfgs:
        if only and name not in only: continue
        try: print(json.dumps(run(name, B, ns, wave)), flush=True)
        except Exception as ex: print(json.dumps(dict(system=name, wave=wave, error=str(ex)[:300])), flush=True)
