"""GPU: one line per BASELINE.json config (C2..C5) with the library's default choices -- RK4
trajectory-steps/s (the metric), the SURVEY-8d HBM fraction, the code path taken, and the reference's
own stepper (stepHam dt) over the same ensemble.  profiles/r01_configs.jsonl is this script's output."""
import os, sys, json
os.environ["HAMK_TEST_OVERRIDES"] = "1"               # HAMK_MAX_SUBSTEPS below is a test override: read only when asked for
os.environ.setdefault("HAMK_MAX_SUBSTEPS", "100000")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamilton_amd import api, examples as E

CONFIGS = [("C2", "doublePendulum", 1 << 20, 100), ("C3", "twoBody", 1 << 20, 100), ("C3", "spring", 1 << 20, 100),
           ("C4", "threeBodyPolar", 1 << 18, 50), ("C5", "chain8", 1 << 16, 50), ("C5", "chain16", 1 << 16, 20),
           ("C5", "chain32", 1 << 16, 10)]

def timed(fn, warm, reps):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

for cfg, name, B, nsteps in CONFIGS:
    spec = E.get(name)
    s = api.system_from_spec(spec)
    q, qd = E.sample_config(spec, 0, B)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    st = api.Phase(ph.positions.clone(), ph.momenta.clone())
    h0 = api.hamiltonian(s, st).clone()
    sec = timed(lambda: api.rk4Steps(spec.dt, nsteps, s, st, inplace=True), 3, 12)
    flagged = int(torch.count_nonzero(s.last_status))
    drift = float(((api.hamiltonian(s, st) - h0).abs() / h0.abs().clamp(min=1.0)).max())
    rate = B * nsteps / sec
    st2 = api.Phase(ph.positions.clone(), ph.momenta.clone())
    hold = [st2]
    def step():
        hold[0] = api.stepHam(spec.dt, s, hold[0], inplace=True)
    sec2 = timed(step, 2, 6)
    nsub = s.last_nsub.double()
    info = {l.split()[0]: l.split()[1] for l in s.build_info.splitlines() if l}
    print(json.dumps(dict(config=cfg, system=name, m=spec.m, n=spec.n, trajectories=B, dt=spec.dt,
                          path="wave" if "INSTANTIATE_WAVE" in s.source else "lane",
                          rk4_steps_per_launch=nsteps, rk4_steps_per_s=rate, hbm_frac_survey_8d=rate * 32 * spec.n / 8e12,
                          rk4_kernel=info["hamk_rk4_steps_k"], status_flagged=flagged, max_rel_energy_drift=drift,
                          stepham_calls_per_s=B / sec2, stepham_mean_substeps=float(nsub.mean()),
                          rkf45_kernel=info["hamk_rkf45_k"])), flush=True)
