"""Pre-compile (hiprtc cross-compiles for gfx950 without a GPU) the code objects the GPU test suite,
bench.py and the sweep scripts will ask for, into a cache directory that travels with the repository
snapshot (`.hamk_cache/`, git-ignored): a fresh GPU box then spends its minutes measuring, not
compiling.  Use on the GPU side with HAMK_CACHE_DIR=$PWD/.hamk_cache.
  python scripts/warm_cache.py [-j 8] [--tests-only]"""
import multiprocessing as mp
import os
os.environ["HAMK_TEST_OVERRIDES"] = "1"               # this script drives libhamk.so through its HAMK_* test overrides
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
CACHE = os.path.join(ROOT, ".hamk_cache")


def jobs():
    out = []
    base = ["pendulum", "doublePendulum", "room", "twoBody", "spring", "bezier", "threeBodyPolar", "opcodeZoo",
            "chain4", "chain5", "chain8", "chain12", "chain16", "chain17", "chain18", "chain20", "chain32", "chain33", "chain40", "chain48", "chain64"]
    for n in base:
        out.append((n, {}, True))
        out.append((n, {"HAMK_GSL_API": "1"}, False))
    for n in ("dense18", "pendulums40", "dense24", "dense32", "chain13", "chain14"):  # round 5: the dense-Jacobian benchmark systems (wave kernels), bench lines with instruction counts
        out.append((n, {}, True))
    # round 6: dense maps on the four-lane kernels (auto: dense18, denseD24; dense24 / dense32 build the quad module, find it spilling and go
    # back to the wave kernels -- both builds are cached), the forced modules of the tests, the wave modules the A/B compares with
    for n in ("denseD24", "denseD32"):
        out.append((n, {}, False))
    for n in ("dense18", "dense24", "dense32", "denseD24", "denseD32"):
        out.append((n, {"HAMK_WAVE": "1"}, False))
    for n in ("dense24", "denseMixed17"):
        out.append((n, {"HAMK_QUAD": "1"}, False))
    for n in ("spring", "threeBodyPolar", "chain4", "opcodeZoo", "chain8", "chain16"):
        out.append((n, {"HAMK_WAVE": "1"}, False))
    for n in ("chain8", "chain16"):
        out.append((n, {"HAMK_WAVE": "0"}, False))
    for n in ("opcodeZoo", "doublePendulum", "spring", "threeBodyPolar"):
        for mode in "HDR":
            for loop in (None, "1"):
                env = {"HAMK_AD_MODE": mode}
                if loop:
                    env["HAMK_RK4_LOOP"] = loop
                out.append((n, env, False))
    for n in ("doublePendulum", "twoBody", "spring", "threeBodyPolar", "pendulum", "chain8", "chain16"):
        out.append((n, {"HAMK_TRIG_LUT": "0"}, False))
        out.append((n, {"HAMK_TRIG_LUT": "1"}, False))
    for seed in range(16):
        out.append((f"random{seed}", {}, False))
    for seed in (0, 2, 3, 5, 6, 7, 8, 11, 13, 14, 15):     # round 6: random trigonometric-polynomial maps (the symbolic mass matrix; tests/test_gpu_random_systems.py)
        out.append((f"polytrig{seed}", {}, False))
    # four lanes per trajectory (hamk_quad.hpp): the default RK4 / hamEqs module of 17 <= n <= 32 (built by the plain jobs
    # above); the wave module those systems keep for their other entry points; forced quad builds of small systems
    for n in ("chain17", "chain18", "chain20", "chain24", "chain32"):
        out.append((n, {"HAMK_WAVE": "1"}, False))
    out.append(("chain24", {}, False))
    for n in ("chain16", "chain8", "threeBodyPolar", "spring", "opcodeZoo"):
        out.append((n, {"HAMK_QUAD": "1"}, False))
    # the adaptive stepper's two lane bodies (stage loop = parked, unrolled) and the quad kernels without parking (tests/test_gpu_wave.py)
    for n in ("chain4", "chain8", "threeBodyPolar", "chain13"):
        out.append((n, {"HAMK_WAVE": "0", "HAMK_QUAD": "0", "HAMK_RKF_LOOP": "1"}, False))
        out.append((n, {"HAMK_WAVE": "0", "HAMK_QUAD": "0", "HAMK_RKF_LOOP": "0"}, False))
    out.append(("chain24", {"HAMK_RKF_PARK": "0"}, False))
    # round 4: systems with a non-positive inertia (lane kernels with the LU fallback, wave kernels with solve_pivoted), the
    # small-ensemble quad module of chain12, the device sampler
    for n in ("doublePendulum~mixed", "spring~mixed", "threeBodyPolar~mixed", "chain6~mixed", "chain12~mixed", "chain20~mixed"):
        out.append((n, {}, False))
    out.append(("chain12", {"HAMK_QUAD": "1"}, False))
    out.append(("sampler", {}, False))
    return out


TESTS_ONLY = "--tests-only" in sys.argv         # tests/conftest.py: no instruction-count probe builds (bench.py's)


def build(job):
    name, env, isa = job
    isa = isa and not TESTS_ONLY
    os.environ["HAMK_CACHE_DIR"] = CACHE
    os.environ.update(env)
    from hamilton_amd import api, examples
    try:
        if name == "sampler":                               # hamk_sample.hpp: compiled before the device is looked at
            spec = examples.get("pendulum")
            s = api.system_from_spec(spec)
            try:
                api.sampleConfig(s, spec.q_box, spec.qd_box, 0, 4, 1)
            except api.HamkError:
                pass
            return name, env, s.code_size
        if name.startswith("random"):
            from test_gpu_random_systems import random_spec
            spec = random_spec(int(name[6:]))
        elif name.startswith("polytrig"):
            from test_gpu_random_systems import poly_trig_spec
            spec = poly_trig_spec(int(name[8:]))
        else:
            spec = examples.get(name)
        s = api.system_from_spec(spec)
        if isa:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import isa_stats
            isa_stats.rk4_step_stats(spec, s)
            if name in ("doublePendulum", "spring", "threeBodyPolar", "twoBody", "chain8", "chain16", "chain32"):
                isa_stats.rkf45_attempt_stats(spec, s)      # bench.py --integrator stepham
        return name, env, s.code_size
    except Exception as e:          # a job that cannot be built here is simply not cached
        return name, env, repr(e)[:200]


if __name__ == "__main__":
    j = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else 8
    os.makedirs(CACHE, mode=0o700, exist_ok=True)
    os.chmod(CACHE, 0o700)
    with mp.Pool(j, maxtasksperchild=1) as pool:
        for name, env, res in pool.imap_unordered(build, jobs()):
            print(name, env, res, flush=True)
