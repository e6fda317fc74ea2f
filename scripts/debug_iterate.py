import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from hamilton_amd import api, examples as E, _abi

spec = E.get("threeBodyPolar")
for park in (_abi.ON, _abi.OFF):
    s = api.system_from_spec(spec, {"rkf_park": park})
    B = 1000
    dt = 3 * spec.dt
    q, qd = E.sample_config(spec, 77, B)
    for dev in (True, False):
        cfg = api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()) if dev else api.Config(q, qd)
        ph0 = api.toPhase(s, cfg)
        for k in (1, 2, 7):
            for every in (0, k):
                a = api.Phase(ph0.positions.clone() if dev else ph0.positions.copy(), ph0.momenta.clone() if dev else ph0.momenta.copy())
                for _ in range(k):
                    a = api.stepHam(dt, s, a)
                r = api.iterateStepHam(dt, k, s, ph0, every=every)
                b = r[0] if every else r
                tonp = lambda x: x.cpu().numpy() if dev else np.asarray(x)
                dq = np.abs(tonp(a.positions) - tonp(b.positions)); dp = np.abs(tonp(a.momenta) - tonp(b.momenta))
                bad = np.nonzero(dp.max(0) > 0)[0]
                print(json.dumps({"park": park, "dev": dev, "k": k, "every": every, "dq": float(dq.max()), "dp": float(dp.max()), "nbad": int(len(bad)), "bad": bad[:8].tolist(),
                                  "rows": np.nonzero(dp.max(1) > 0)[0].tolist()}), flush=True)
