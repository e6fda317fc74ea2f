"""GPU: time attribution of the wave-cooperative RHS (chain32) by removing one phase at a time
(HAMK_PROBE_SKIP_* macros in hamk_wave.hpp; results are wrong by construction, only the time counts)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["HAMK_SELFCHECK"] = "0"
from sweep_wave import run
name = sys.argv[1] if len(sys.argv) > 1 else "chain32"
variants = [("full", ""), ("-kacc", "-DHAMK_PROBE_SKIP_KACC"), ("-factor", "-DHAMK_PROBE_SKIP_FACTOR"),
            ("-solve", "-DHAMK_PROBE_SKIP_SOLVE"), ("-sweep2", "-DHAMK_PROBE_SKIP_SWEEP2"),
            ("-kacc-factor", "-DHAMK_PROBE_SKIP_KACC -DHAMK_PROBE_SKIP_FACTOR"),
            ("-all4", "-DHAMK_PROBE_SKIP_KACC -DHAMK_PROBE_SKIP_FACTOR -DHAMK_PROBE_SKIP_SOLVE -DHAMK_PROBE_SKIP_SWEEP2")]
for label, flags in variants:
    if flags: os.environ["HAMK_HIPRTC_FLAGS"] = flags
    else: os.environ.pop("HAMK_HIPRTC_FLAGS", None)
    try:
        r = run(name, 65536, 10, None)
        print(json.dumps(dict(variant=label, ms=r["ms"], steps_per_s=r["steps_per_s"], compile_s=r["compile_s"])), flush=True)
    except Exception as ex:
        print(json.dumps(dict(variant=label, error=str(ex)[:300])), flush=True)
