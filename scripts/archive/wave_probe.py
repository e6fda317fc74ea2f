"""GPU: two launches of the wave-cooperative RK4 kernel (B = 65536, 4 steps) -- the command rocprofv3's
kernel-trace / PMC passes are wrapped around (profiles/r01_wave_pmc.json, r01_wave_kernel_stats.csv)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamilton_amd import api, examples as E
name = sys.argv[1] if len(sys.argv) > 1 else "chain32"
spec = E.get(name); s = api.system_from_spec(spec)
B = 65536
q, qd = E.sample_config(spec, 0, B)
ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
st = api.Phase(ph.positions.clone(), ph.momenta.clone())
for _ in range(2):
    api.rk4Steps(spec.dt, 4, s, st, inplace=True)
torch.cuda.synchronize()
