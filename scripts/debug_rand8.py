import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from hamilton_amd import api, examples as E
from oracle import oracle
from test_gpu_random_systems import random_spec
mode = sys.argv[1]
seeds = [int(a) for a in sys.argv[2:]] or [8]
for seed in seeds:
    spec = random_spec(seed)
    s = api.system_from_spec(spec); o = oracle.OracleSystem(spec)
    B = 130
    q, qd = E.sample_config(spec, 99, B)
    p = o.to_phase_batch(q, qd)
    sq, sp, sns = o.step_ham_batch(q, p, 0.02)
    if os.environ.get("PRE", "1") == "1":
        api.hamEqs(s, api.Phase(q, p)); api.rk4Steps(spec.dt, 2, s, api.Phase(q, p))
    res = []
    for rep in range(6):
        if mode == "dev":
            st = api.stepHam(0.02, s, api.Phase(torch.from_numpy(q).cuda(), torch.from_numpy(p).cuda()))
            torch.cuda.synchronize()
            res.append((s.last_nsub.cpu().numpy().copy(), st.positions.cpu().numpy().copy()))
        else:
            st = api.stepHam(0.02, s, api.Phase(q, p)); res.append((np.asarray(s.last_nsub).copy(), st.positions.copy()))
    eq = [bool((r[0] == res[0][0]).all() and (r[1] == res[0][1]).all()) for r in res]
    print(mode, "PINNED=" + os.environ.get("HAMK_PINNED", "1"), "seed", seed, "n", spec.n, "m", spec.m, "gpu==oracle per rep",
          [round(float((r[0] == sns).mean()), 3) for r in res], "reps equal", eq, flush=True)
    for r in res:
        i = np.nonzero(r[0] != sns)[0]
        if len(i): print("   lanes", i[:12], "gpu", r[0][i[:12]], "maxdq", float(np.abs(r[1][:, i] - sq[:, i]).max()))
    print("   source RKF_STAGE_LOOP:", "RKF_STAGE_LOOP = true" in s.source, " NTRIG_F", [l for l in s.source.splitlines() if "NTRIG_F =" in l][:1])
