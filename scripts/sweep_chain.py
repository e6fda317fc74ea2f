"""GPU: the fixed-step kernels with sincos through the LDS table (hamk_device.hpp sincos_lut, the
default) against the anchor scheme it replaces (HAMK_TRIG_LUT=0: full evaluation + rotations about the
step's midpoint for 1-4 sincos sites, full evaluations beyond): RK4 steps/s at BASELINE ensemble size.
Output: one JSON line per measurement (profiles/r02_sweep_trig.jsonl)."""
import json, os, sys
os.environ["HAMK_TEST_OVERRIDES"] = "1"               # drives libhamk.so through its HAMK_* test overrides
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamilton_amd import api, examples as E


def timed(fn, warm, reps):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def run(name, B, nsteps, env, drift_tol=0.0, reps=8):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        spec = E.get(name)
        s = api.system_from_spec(spec)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    q, qd = E.sample_config(spec, 0, B)
    ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    st = api.Phase(ph.positions.clone(), ph.momenta.clone())
    sec = timed(lambda: api.rk4Steps(spec.dt, nsteps, s, st, inplace=True, drift_tol=drift_tol), 3, reps)
    info = {l.split()[0]: " ".join(l.split()[1:]) for l in s.build_info.splitlines() if l}
    return dict(system=name, B=B, rk4_steps_per_launch=nsteps, env=env, drift_tol=drift_tol, ms=sec * 1e3,
                steps_per_s=B * nsteps / sec, rk4_kernel=info["hamk_rk4_steps_k"])


def variants():
    out = []
    for name, B, ns in (("doublePendulum", 1 << 20, 400), ("twoBody", 1 << 20, 400), ("spring", 1 << 20, 400), ("threeBodyPolar", 1 << 18, 400),
                        ("pendulum", 1 << 20, 400), ("chain8", 1 << 16, 200), ("chain16", 1 << 16, 50)):
        out.append((name, B, ns, {"HAMK_TRIG_LUT": "2"}, 0.0, 10))            # default: table for the full evaluation, rotations (1-4 sites)
        out.append((name, B, ns, {"HAMK_TRIG_LUT": "1"}, 0.0, 10))            # every sincos through the LDS table
        out.append((name, B, ns, {"HAMK_TRIG_LUT": "0"}, 0.0, 10))            # no table: round-1 arithmetic + midpoint anchors
    for name, ns in (("chain16", 50), ("chain12", 100)):                      # register-bound lane kernels: which build, table or not
        for lut in ("1", "0"):
            for nolicm in ("0", "1"):
                out.append((name, 1 << 16, ns, {"HAMK_TRIG_LUT": lut, "HAMK_NOLICM": nolicm}, 0.0, 8))
    return out


if __name__ == "__main__":
    if "--warm" in sys.argv:              # no GPU: only compile the variants into $HAMK_CACHE_DIR
        for name, B, nsteps, env, tol, reps in variants():
            os.environ.update(env)
            api.system_from_spec(E.get(name))
            for k in env: os.environ.pop(k, None)
            print("built", name, env, flush=True)
        sys.exit(0)
    for name, B, nsteps, env, tol, reps in variants():
        print(json.dumps(run(name, B, nsteps, env, drift_tol=tol, reps=reps)), flush=True)
