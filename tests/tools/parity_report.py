"""GPU: one table of measured parity (HIP path through the C ABI vs the CPU oracle and vs the
high-precision fixtures) per system -- the evidence behind the tolerance ladder of DESIGN.md section 5."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hamilton_amd import api, examples as E
from oracle import oracle

def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(1.0, np.abs(np.asarray(b)))))

def fv(xs):
    return np.array([float(x) for x in xs])

rows = []
for name in ["pendulum", "doublePendulum", "room", "twoBody", "spring", "bezier", "threeBodyPolar", "chain4", "opcodeZoo",
             "chain8", "chain12", "chain16", "chain20", "chain32"]:
    spec = E.get(name); s = api.system_from_spec(spec); o = oracle.OracleSystem(spec)
    B = 2048 if spec.n <= 8 else 256
    q, qd = E.sample_config(spec, 4242, B)
    if name.startswith("chain"):
        qd = 0.3 * np.cos(np.arange(spec.n * B).reshape(spec.n, B) * 0.7)
    p = o.to_phase_batch(q, qd)
    dq, dp = api.hamEqs(s, api.Phase(q, p)); odq, odp, _ = o.hameqs_batch(q, p)
    r = {"system": name, "m": spec.m, "n": spec.n, "path": "wave" if "INSTANTIATE_WAVE" in s.source else "lane", "B": B,
         "toPhase": rel(api.momenta(s, api.Config(q, qd)), p), "hamEqs": max(rel(dq, odq), rel(dp, odp)),
         "hamiltonian": rel(api.hamiltonian(s, api.Phase(q, p)), o.observe_batch(q, p)[2])}
    one = api.rk4Steps(spec.dt, 1, s, api.Phase(q, p)); oq, op = o.rk4_steps_batch(q, p, spec.dt, 1)
    r["rk4_1_step"] = max(rel(one.positions, oq), rel(one.momenta, op))
    many = api.rk4Steps(spec.dt, 100, s, api.Phase(q, p)); oq, op = o.rk4_steps_batch(q, p, spec.dt, 100)
    r["rk4_100_steps"] = max(rel(many.positions, oq), rel(many.momenta, op))
    st = api.stepHam(spec.dt, s, api.Phase(q, p)); sq, sp, sns = o.step_ham_batch(q, p, spec.dt)
    same = np.asarray(s.last_nsub) == sns
    r["stepHam_same_substeps"] = float(same.mean())
    r["stepHam"] = max(rel(st.positions[:, same], sq[:, same]), rel(st.momenta[:, same], sp[:, same]))
    ts = np.array([0.0, 3 * spec.dt, 7 * spec.dt, 7 * spec.dt, 12 * spec.dt])
    for gsl in (2, 1):                                   # evolveHam under both bindings of hmatrix-gsl's gsl-ode.c
        s.gsl_api = gsl; o.gsl_api = gsl
        rows_ = api.evolveHam(s, api.Phase(q, p), ts); eq_, ep_, ens = o.evolve_ham_batch(q, p, ts)
        same = np.asarray(s.last_nsub) == ens
        r[f"evolveHam_api{gsl}_same_substeps"] = float(same.mean())
        r[f"evolveHam_api{gsl}"] = max(max(rel(rows_[k].positions[:, same], eq_[k][:, same]), rel(rows_[k].momenta[:, same], ep_[k][:, same])) for k in range(1, len(ts)))
    s.gsl_api = 2; o.gsl_api = 2
    gpath = os.path.join(ROOT, "tests", "golden", f"{name}.json")
    if os.path.exists(gpath):
        pts = json.load(open(gpath))["points"]
        gq = np.stack([fv(t["q"]) for t in pts], 1); gp = np.stack([fv(t["p"]) for t in pts], 1)
        gdq, gdp = api.hamEqs(s, api.Phase(gq, gp))
        r["hamEqs_vs_50digit_fixture"] = max(rel(gdq, np.stack([fv(t["dq"]) for t in pts], 1)), rel(gdp, np.stack([fv(t["dp"]) for t in pts], 1)))
    rows.append(r)
    print(json.dumps(r), flush=True)
