"""GPU: the wave-cooperative RK4 kernel at 1, 2 (default) and 3 wavefronts per SIMD (HAMK_RK4_WAVES)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_wave import run
for waves in ("3", "1", ""):
    if waves: os.environ["HAMK_RK4_WAVES"] = waves
    else: os.environ.pop("HAMK_RK4_WAVES", None)
    for name, ns in (("chain32", 10), ("chain20", 10)):
        r = run(name, 65536, ns, None)
        print("waves", waves or "default(2)", name, "%.3e" % r["steps_per_s"], flush=True)
