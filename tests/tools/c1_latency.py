"""BASELINE config 1 ("plumbing"): double pendulum, ONE trajectory from seInit, 1000 x stepHam 0.01.
Latency of the host-staged path per call vs the CPU oracle (the GPU is latency-bound at B = 1)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hamilton_amd import api, examples as E
from oracle import oracle
spec = E.get("doublePendulum"); s = api.system_from_spec(spec); o = oracle.OracleSystem(spec)
q, p = np.array(spec.q0), np.zeros(2)
ph = api.stepHam(0.01, s, api.Phase(q, p))      # warm-up (module load)
t0 = time.perf_counter()
for _ in range(1000):
    ph = api.stepHam(0.01, s, api.Phase(q, p)); q, p = ph.positions, ph.momenta
g = time.perf_counter() - t0
# the same through the bare C ABI (no Python object churn): what a compiled host pays per call
import ctypes
from hamilton_amd import _abi
lib = _abi.lib()
rq, rp = np.array(spec.q0), np.zeros(2); st = np.zeros(1, np.int32); ns = np.zeros(1, np.int32)
dp = ctypes.POINTER(ctypes.c_double); ip = ctypes.POINTER(ctypes.c_int32)
args = (s._h, 1, rq.ctypes.data_as(dp), rp.ctypes.data_as(dp), ctypes.c_double(0.01), st.ctypes.data_as(ip), ns.ctypes.data_as(ip), 0)
t0 = time.perf_counter()
for _ in range(1000):
    lib.hamk_step_ham_batch(*args)
raw = time.perf_counter() - t0
assert max(np.max(np.abs(rq - q)), np.max(np.abs(rp - p))) == 0.0
print(f"C1 bare C ABI: {raw*1e3:.1f} us/call (HAMK_PINNED={os.environ.get('HAMK_PINNED', '1')})")
oq, op = np.array(spec.q0), np.zeros(2)
t0 = time.perf_counter()
for _ in range(1000):
    oq, op = o.step_ham(0.01, oq, op)
c = time.perf_counter() - t0
r4 = api.rk4Steps(0.01, 1000, s, api.Phase(np.array(spec.q0), np.zeros(2)))
print(f"C1 stepHam x1000: GPU path {g*1e3:.1f} ms ({g*1e3:.1f} us/call), CPU oracle {c*1e3:.1f} ms ({c*1e3:.1f} us/call); "
      f"max|dphase| GPU vs oracle {max(np.max(np.abs(q-oq)), np.max(np.abs(p-op))):.2e}; "
      f"RK4 x1000 vs stepHam x1000 (truncation): {max(np.max(np.abs(r4.positions-q)), np.max(np.abs(r4.momenta-p))):.2e}")
