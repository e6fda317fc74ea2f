"""GPU: does a register cap (more wavefronts per SIMD, __launch_bounds__(256, k) through HAMK_RK4_WAVES) pay
for the VALU-bound lane kernels?  And the reference's own stepper over the ensemble (stepHam dt).
Output: one JSON line per measurement (profiles/r02_sweep_occupancy.jsonl)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hamilton_amd import api, examples as E
from sweep_chain import run, timed


def variants():
    out = []
    for name, B, ns in (("doublePendulum", 1 << 20, 400), ("twoBody", 1 << 20, 400), ("spring", 1 << 20, 400), ("threeBodyPolar", 1 << 18, 400)):
        for k in (None, "2", "3", "4", "5"):
            out.append((name, B, ns, {} if k is None else {"HAMK_RK4_WAVES": k}))
    return out


def stepham(name, B):
    spec = E.get(name)
    s = api.system_from_spec(spec)
    q, qd = E.sample_config(spec, 0, B)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    st = api.Phase(ph.positions.clone(), ph.momenta.clone())
    sec = timed(lambda: api.stepHam(spec.dt, s, st, inplace=True), 3, 10)
    ns = s.last_nsub.double()
    return dict(system=name, B=B, integrator="stepHam dt", dt=spec.dt, calls_per_s=B / sec, ms=sec * 1e3,
                mean_substeps=float(ns.mean()), max_substeps=float(ns.max()), gsl_api=s.gsl_api)


if __name__ == "__main__":
    if "--warm" in sys.argv:
        for name, B, ns, env in variants():
            os.environ.update(env)
            api.system_from_spec(E.get(name))
            for k in env: os.environ.pop(k, None)
            print("built", name, env, flush=True)
        sys.exit(0)
    for name, B, ns, env in variants():
        try:
            print(json.dumps(run(name, B, ns, env, reps=8)), flush=True)
        except Exception as ex:
            print(json.dumps(dict(system=name, env=env, error=str(ex)[:200])), flush=True)
    for name, B in (("doublePendulum", 1 << 20), ("twoBody", 1 << 20), ("spring", 1 << 20), ("threeBodyPolar", 1 << 18), ("chain8", 1 << 16), ("chain32", 1 << 16)):
        print(json.dumps(stepham(name, B)), flush=True)
