"""GPU: the PCIe-inclusive rate of the headline workload -- config 2 handed over as HOST arrays
(HAMK_MEM_HOST: staged copies in, 100 fused RK4 steps, copies out, per call).  Never bench.py's `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamilton_amd import api, examples as E
spec = E.get("doublePendulum"); s = api.system_from_spec(spec)
B = 1 << 20
q, qd = E.sample_config(spec, 0, B)
ph = api.toPhase(s, api.Config(q, qd))
for nsteps in (1, 10, 100, 1000):
    api.rk4Steps(spec.dt, nsteps, s, ph)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): out = api.rk4Steps(spec.dt, nsteps, s, ph)
    el = (time.perf_counter() - t0) / reps
    print(f"host arrays, {nsteps:5d} fused steps per call: {el*1e3:8.2f} ms/call  {B*nsteps/el:.3e} trajectory-steps/s "
          f"({64.0*B/el/1e9:.1f} GB/s of state over PCIe incl. staging)", flush=True)
