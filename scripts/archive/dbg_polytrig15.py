"""Round 6 debug: which lanes of polytrig15 differ from the oracle after 4 RK4 steps on the GPU, and what they look like."""
import os, sys, json, numpy as np
os.environ["HAMK_TEST_OVERRIDES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hamilton_amd import api, examples as E
from oracle import oracle as O
from test_gpu_random_systems import poly_trig_spec
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 15
spec = poly_trig_spec(seed)
s = api.system_from_spec(spec)
o = O.OracleSystem(spec)
B = 257
q, qd = E.sample_config(spec, 5, B)
p = o.to_phase_batch(q, qd)
odq, odp, ost = o.hameqs_batch(q, p)
cond = np.array([np.linalg.cond(o.jacobian(q[:, i]).T @ np.diag(spec.inertia) @ o.jacobian(q[:, i])) for i in range(B)])
for nsteps in (1, 2, 4):
    ph = api.rk4Steps(spec.dt, nsteps, s, api.Phase(q, p))
    st = np.asarray(s.last_status)
    oq, op = o.rk4_steps_batch(q, p, spec.dt, nsteps)
    gq, gp = np.asarray(ph.positions), np.asarray(ph.momenta)
    err = np.maximum(np.abs(gq - oq).max(0) / np.maximum(1, np.abs(oq).max(0)), np.abs(gp - op).max(0) / np.maximum(1, np.abs(op).max(0)))
    bad = np.where(err > 1e-9)[0]
    print("nsteps", nsteps, "bad", bad.tolist(), "err", err[bad].tolist(), "cond0", cond[bad].tolist(), "status", st[bad].tolist(), "ost", ost[bad].tolist())
    for i in bad[:4]:
        print("  lane", i, "q0", q[:, i].tolist(), "p0", p[:, i].tolist())
        print("     gpu q", gq[:, i].tolist(), "p", gp[:, i].tolist())
        print("     ora q", oq[:, i].tolist(), "p", op[:, i].tolist())
# one step at a time from the oracle's states: where does a single step differ?
qq, pp = q.copy(), p.copy()
for k in range(4):
    ph = api.rk4Steps(spec.dt, 1, s, api.Phase(qq, pp))
    oq, op = o.rk4_steps_batch(qq, pp, spec.dt, 1)
    gq, gp = np.asarray(ph.positions), np.asarray(ph.momenta)
    err = np.maximum(np.abs(gq - oq).max(0), np.abs(gp - op).max(0))
    bad = np.where(err > 1e-9)[0]
    print("single step", k, "bad", bad.tolist(), err[bad].tolist())
    dq, dp = api.hamEqs(s, api.Phase(qq, pp)); odq, odp, _ = o.hameqs_batch(qq, pp)
    e2 = np.maximum(np.abs(np.asarray(dq) - odq).max(0), np.abs(np.asarray(dp) - odp).max(0))
    print("   hamEqs at these states: max err", e2.max(), "at lane", int(e2.argmax()), "cond there", float(np.linalg.cond(o.jacobian(qq[:, int(e2.argmax())]).T @ np.diag(spec.inertia) @ o.jacobian(qq[:, int(e2.argmax())]))))
    qq, pp = oq, op
print([l for l in s.build_info.splitlines() if "rk4" in l])
