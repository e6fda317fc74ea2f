"""Experiment (round 6): which fences the burst of table gathers needs -- RK4 lane kernels of the mid-size chains, where the shipped form loses.
Ran against a COPY of the package (exp/hamilton_amd) whose trig_burst_lut took its fence mask and its second fence from macros
(HAMK_TRIG_BURST_MASK, HAMK_TRIG_BURST_FENCE2); none of the forms moves the mid-size kernels, the copy was deleted."""
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
VARIANTS = (("shipped rule", ""), ("gathers ahead, rotations in the sweep: where the burst is off", "-DHAMK_TRIG_PREFETCH=1"), ("gathers ahead, rotations in the sweep: everywhere", "-DHAMK_TRIG_PREFETCH=2"))
SYSTEMS = (("chain6", 400), ("chain7", 400), ("chain8", 400), ("chain9", 300), ("chain10", 300), ("chain11", 200), ("chain12", 200), ("chain13", 200), ("chain14", 100), ("chain16", 100))


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
    states = [api.Phase(ph.positions.clone(), ph.momenta.clone()) for _ in built]
    best = [None] * len(built)
    for rnd in range(3):
        for i, (tag, flags, s) in enumerate(built):
            for _ in range(5):
                ms = timed(lambda: api.rk4Steps(spec.dt, nsteps, s, states[i], inplace=True))
                best[i] = ms if best[i] is None else min(best[i], ms)
    for i, (tag, flags, s) in enumerate(built):
        print(json.dumps({"what": "trig_prefetch_ab", "system": name, "B": B, "variant": tag, "flags": flags, "rk4_steps_per_s": B * nsteps / (best[i] * 1e-3),
                          "vs_off": best[0] / best[i]}), flush=True)
