"""GPU: AD strategy (HAMK_AD_MODE = H full second-order jets / D directional second sweep / R reverse second
sweep) x RK4 body (unrolled / stage loop) for the mid-size systems, with the round-2 sincos in place --
the thresholds in hamk_system_create (H up to n = 3, R from n = 8, stage loop from n = 7) were measured in
round 1.  Output: one JSON line per variant (profiles/r02_sweep_modes.jsonl)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hamilton_amd import api, examples as E


def variants():
    out = []
    for name, B, ns in (("spring", 1 << 20, 200), ("threeBodyPolar", 1 << 18, 200), ("chain4", 1 << 18, 200), ("chain8", 1 << 16, 100)):
        for mode in "HDR":
            if name == "chain8" and mode == "H":
                continue
            for loop in ("0", "1"):
                out.append((name, B, ns, {"HAMK_AD_MODE": mode, "HAMK_RK4_LOOP": loop}))
        out.append((name, B, ns, {}))
    return out


if __name__ == "__main__":
    if "--warm" in sys.argv:
        for name, B, ns, env in variants():
            os.environ.update(env)
            try:
                api.system_from_spec(E.get(name))
                print("built", name, env, flush=True)
            except Exception as ex:
                print("FAILED", name, env, str(ex)[:100], flush=True)
            for k in env: os.environ.pop(k, None)
        sys.exit(0)
    from sweep_chain import run
    for name, B, ns, env in variants():
        try:
            print(json.dumps(run(name, B, ns, env, reps=8)), flush=True)
        except Exception as ex:
            print(json.dumps(dict(system=name, env=env, error=str(ex)[:200])), flush=True)
