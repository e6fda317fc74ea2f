"""GPU: is the adaptive stepper of a system deterministic, and does the unrolled body agree with
the stage-loop body?  (The unrolled RKF45 kernel of a random test system was found to give
run-to-run different, wrong results on some lanes: a code-generation hazard, not a data race --
every lane is independent.)   python scripts/determinism.py <system|randomK> [B] [reps]"""
import os, sys
os.environ["HAMK_TEST_OVERRIDES"] = "1"               # this script drives libhamk.so through its HAMK_* test overrides
os.environ.setdefault("HAMK_MAX_SUBSTEPS", "2000")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hamilton_amd import api, examples as E
name = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if name.startswith("random"):
    from test_gpu_random_systems import random_spec
    spec = random_spec(int(name[6:]))
else:
    spec = E.get(name)
dt = float(os.environ.get("DT", "0.02"))
def build(loop):
    if loop is None: os.environ.pop("HAMK_RKF_LOOP", None)
    else: os.environ["HAMK_RKF_LOOP"] = loop
    return api.system_from_spec(spec)
s_def, s_loop = build(os.environ.get("DEF_LOOP")), build("1")
q, qd = E.sample_config(spec, 99, B)
tq, tqd = torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()
ph = api.toPhase(s_loop, api.Config(tq, tqd))
api.hamEqs(s_def, ph); api.rk4Steps(spec.dt, 2, s_def, ph)      # other kernels first: they leave other register contents behind
def run(s):
    out = []
    for _ in range(reps):
        st = api.stepHam(dt, s, ph); torch.cuda.synchronize()
        out.append((st.positions.clone(), st.momenta.clone(), s.last_nsub.clone(), s.last_status.clone()))
    return out
a, b = run(s_def), run(s_loop)
det = lambda r: [bool(torch.equal(x[0], r[0][0]) and torch.equal(x[1], r[0][1]) and torch.equal(x[2], r[0][2])) for x in r]
ok = (b[0][3] == 0)
nsame = [float((x[2] == b[0][2])[ok].double().mean()) for x in a]
dmax = [float(((x[0] - b[0][0]).abs().max(0).values[ok & (x[2] == b[0][2])]).max()) for x in a]
print(name, "B", B, "default body is stage loop:", "RKF_STAGE_LOOP = true" in s_def.source,
      "| kernel bytes", s_def.kernel_bytes("hamk_rkf45_k"), "vs loop", s_loop.kernel_bytes("hamk_rkf45_k"))
print("  default deterministic:", det(a), " loop deterministic:", det(b))
print("  default vs loop: nsub equal fraction per rep", [round(v, 4) for v in nsame], " max|dq| where equal", ["%.1e" % v for v in dmax])
