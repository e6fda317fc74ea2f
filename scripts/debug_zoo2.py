import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamilton_amd import api, examples as E
from oracle import oracle
spec = E.get("opcodeZoo"); o = oracle.OracleSystem(spec)
B = 257
q, qd = E.sample_config(spec, 2024, B)
p = o.to_phase_batch(q, qd)
for mode, loop, flags, waves in (("H", "0", "", ""), ("H", "0", "-fhonor-nans -fsigned-zeros", ""), ("H", "0", "-O1", ""), ("H", "0", "-O2", ""),
                                 ("H", "0", "-mllvm -amdgpu-use-divergent-register-indexing", ""), ("H", "0", "-mllvm -amdgpu-spill-sgpr-to-vgpr=0", ""),
                                 ("H", "0", "-mllvm -amdgpu-spill-vgpr-to-agpr=0", "")):
    os.environ["HAMK_AD_MODE"] = mode; os.environ["HAMK_RK4_LOOP"] = loop
    os.environ["HAMK_HIPRTC_FLAGS"] = flags
    if waves: os.environ["HAMK_RK4_WAVES"] = waves
    else: os.environ.pop("HAMK_RK4_WAVES", None)
    print("flags", flags, "waves", waves)
    s = api.system_from_spec(spec)
    for n in (1, 2, 3):
        ph = api.rk4Steps(spec.dt, n, s, api.Phase(q, p))
        oq, op = o.rk4_steps_batch(q, p, spec.dt, n)
        err = np.maximum(np.abs(ph.positions - oq).max(0), np.abs(ph.momenta - op).max(0))
        bad = np.where(err > 1e-9)[0]
        print(mode, loop, "steps", n, "max err %.2e" % err.max(), "bad lanes", bad[:10], "status", np.asarray(s.last_status)[bad[:10]], flush=True)
    if len(bad):
        i = bad[0]
        print(" lane", i, "q", q[:, i], "p", p[:, i], "oracle cond:", np.linalg.cond((lambda J: J.T @ np.diag(spec.inertia) @ J)(o.jacobian(q[:, i]))))
