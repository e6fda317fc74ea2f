"""Experiment (round 6): the parked RK4 kernels' stage boundary in one LDS round trip (HAMK_PARK_BURST) -- chain14 ... chain16, lane kernels.
Ran against a COPY of the package (exp/hamilton_amd) whose parked RK4 loop loaded the rows of the combination (and, form 1, of the next
base point) back to back behind a scheduling fence; the copy was deleted after the measurement."""
import json, os, sys
os.environ["HAMK_TEST_OVERRIDES"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
os.environ["HAMK_CACHE_DIR"] = os.path.join(HERE, ".hamk_cache")
os.makedirs(os.environ["HAMK_CACHE_DIR"], mode=0o700, exist_ok=True); os.chmod(os.environ["HAMK_CACHE_DIR"], 0o700)
COMPILE_ONLY = "--compile-only" in sys.argv
from hamilton_amd import _abi, api, examples
assert api.__file__.startswith(HERE), api.__file__
if not COMPILE_ONLY:
    import torch
VARIANTS = (("a round trip per few components", "-DHAMK_PARK_BURST=0"), ("one round trip per stage boundary", ""), ("two round trips per stage boundary", "-DHAMK_PARK_BURST=2"))
SYSTEMS = (("chain14", 100), ("chain15", 100), ("chain16", 100))


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


B = 1 << 16
for name, nsteps in SYSTEMS:
    spec = examples.get(name)
    built = []
    for tag, flags in VARIANTS:
        os.environ["HAMK_HIPRTC_FLAGS"] = flags
        s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE})
        if COMPILE_ONLY:
            print(name, tag, [l for l in s.build_info.splitlines() if l.startswith("hamk_rk4_steps_k")], flush=True)
        built.append((tag, flags, s))
    if COMPILE_ONLY:
        continue
    q, qd = examples.sample_config(spec, 0, B)
    ph = api.toPhase(built[0][2], api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
    outs = [api.rk4Steps(spec.dt, 20, s, api.Phase(ph.positions.clone(), ph.momenta.clone())) for _, _, s in built]
    same = all(bool(torch.equal(outs[0].positions, o.positions) and torch.equal(outs[0].momenta, o.momenta)) for o in outs[1:])
    states = [api.Phase(ph.positions.clone(), ph.momenta.clone()) for _ in built]
    best = [None] * len(built)
    for rnd in range(3):
        for i, (tag, flags, s) in enumerate(built):
            for _ in range(5):
                ms = timed(lambda: api.rk4Steps(spec.dt, nsteps, s, states[i], inplace=True))
                best[i] = ms if best[i] is None else min(best[i], ms)
    for i, (tag, flags, s) in enumerate(built):
        print(json.dumps({"what": "park_burst_ab", "system": name, "B": B, "variant": tag, "flags": flags, "rk4_steps_per_s": B * nsteps / (best[i] * 1e-3),
                          "vs_first": best[0] / best[i], "bitwise_equal_after_20_steps": same}), flush=True)
