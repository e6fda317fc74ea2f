#!/usr/bin/env python
"""BASELINE config 1 as a latency: one trajectory, stepHam 0.01 call after call through the host-pointer path of the C ABI
(pinned arena: one launch + one wait per call), microseconds per call; the Python mirror and the bare ctypes call."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamilton_amd import _abi, api, examples  # noqa: E402

spec = examples.get("doublePendulum")
s = api.system_from_spec(spec)
ph = api.toPhase(s, api.Config(np.array(spec.q0, dtype=np.float64), np.array(spec.qd0, dtype=np.float64)))
q, p = np.asarray(ph.positions, dtype=np.float64).copy(), np.asarray(ph.momenta, dtype=np.float64).copy()
for _ in range(50):
    api.stepHam(0.01, s, api.Phase(q, p))
best = None
for _ in range(5):
    gq, gp = q, p
    t0 = time.perf_counter()
    for _ in range(1000):
        r = api.stepHam(0.01, s, api.Phase(gq, gp))
        gq, gp = r.positions, r.momenta
    t = (time.perf_counter() - t0) * 1e3
    best = t if best is None else min(best, t)
# the bare C call (what a compiled host pays)
lib = _abi.lib()
cq, cp = q.copy().reshape(spec.n, 1), p.copy().reshape(spec.n, 1)
st = np.zeros(1, dtype=np.int32)
ns = np.zeros(1, dtype=np.int32)
aq, ap, ast_, ans = cq.ctypes.data, cp.ctypes.data, st.ctypes.data, ns.ctypes.data
call = lambda: lib.hamk_step_ham_batch(s._h, 1, aq, ap, 0.01, ast_, ans, _abi.MEM_HOST)
for _ in range(50):
    assert call() == 0
bare = None
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(1000):
        call()
    t = (time.perf_counter() - t0) * 1e3
    bare = t if bare is None else min(bare, t)
print(json.dumps({"us_per_call_python_mirror": best, "us_per_call_bare_ctypes": bare, "spin": os.environ.get("HAMK_SPIN", "default")}))
