import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamilton_amd import api, examples as E
from oracle import oracle
for name in ("opcodeZoo", "threeBodyPolar", "chain8", "doublePendulum"):
    spec = E.get(name); o = oracle.OracleSystem(spec)
    B = 1 << 14
    q, qd = E.sample_config(spec, 2024, B)
    p = o.to_phase_batch(q, qd)
    sq, sp, sns = o.step_ham_batch(q, p, 0.01)
    for loop in ("0", "1"):
        os.environ["HAMK_RKF_LOOP"] = loop
        t0 = time.time(); s = api.system_from_spec(spec); tc = time.time() - t0
        st = api.stepHam(0.01, s, api.Phase(q, p))
        t0 = time.time()
        for _ in range(5): st = api.stepHam(0.01, s, api.Phase(q, p))
        el = (time.time() - t0) / 5
        ns = np.asarray(s.last_nsub)
        print(name, "rkf_loop", loop, "compile %.1fs" % tc, "nsub match %.3f" % (ns == sns).mean(), "gpu nsub[:4]", ns[:4], "oracle", sns[:4],
              "maxdiff %.2e" % max(np.max(np.abs(st.positions - sq)), np.max(np.abs(st.momenta - sp))), "host-call ms %.2f" % (el * 1e3), flush=True)
