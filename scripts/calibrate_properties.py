"""GPU diagnostic: distribution of RK4 time-reversal error and energy drift over the full
config-2 ensemble (used to set the thresholds of tests/test_gpu_parity.py::test_full_size_properties)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamilton_amd import api, examples as E

spec = E.get("doublePendulum"); s = api.system_from_spec(spec)
B = 1 << 20
q, qd = E.sample_config(spec, 0, B)
ph0 = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
h0 = api.hamiltonian(s, ph0)
qs = torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], dtype=torch.float64, device="cuda")
for dt, n in ((0.01, 100), (0.005, 200), (0.0025, 400)):
    ph1 = api.rk4Steps(dt, n, s, ph0)
    back = api.rk4Steps(-dt, n, s, ph1)
    err = torch.maximum((back.positions - ph0.positions).abs().amax(0), (back.momenta - ph0.momenta).abs().amax(0))
    drift = (api.hamiltonian(s, ph1) - h0).abs() / h0.abs().clamp(min=1.0)
    print(f"dt={dt} n={n} reversal quantiles {torch.quantile(err, qs).tolist()} mean {float(err.mean()):.3e}")
    print(f"           drift quantiles    {torch.quantile(drift, qs).tolist()} mean {float(drift.mean()):.3e}")
