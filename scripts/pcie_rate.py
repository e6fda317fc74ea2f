#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer path for large ensembles (never bench.py's `value`): one RK4 step of the config-2
ensemble handed over as numpy arrays, 2 x 2 x 8 B x B in and out per call, with the pinned bounce buffers + parallel
memcpy (default) and with plain hipMemcpy from pageable memory (HAMK_BOUNCE=0):
  python scripts/pcie_rate.py            (runs both in sub-processes)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one():
    sys.path.insert(0, ROOT)
    import numpy as np
    from hamilton_amd import api, examples
    spec = examples.get("doublePendulum")
    s = api.system_from_spec(spec)
    for B in (1 << 20, 1 << 22):
        q, qd = examples.sample_config(spec, 0, B)
        ph = api.toPhase(s, api.Config(q, qd))
        api.rk4Steps(0.01, 1, s, ph)
        best = None
        for _ in range(5):
            t0 = time.perf_counter()
            api.rk4Steps(0.01, 1, s, ph)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        moved = 2 * 2 * spec.n * B * 8
        print(json.dumps({"bounce": os.environ.get("HAMK_BOUNCE", "1"), "threads": os.environ.get("HAMK_COPY_THREADS", "8"), "B": B, "call_ms": best * 1e3,
                          "bytes_each_way": moved // 2, "GB_per_s_both_ways": moved / best / 1e9, "trajectory_steps_per_s": B / best}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for env in ({"HAMK_BOUNCE": "0"}, {}, {"HAMK_COPY_THREADS": "1"}, {"HAMK_COPY_THREADS": "4"}):
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **env), check=False)
