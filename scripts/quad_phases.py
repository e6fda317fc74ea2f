#!/usr/bin/env python3
"""Where the instructions of one right-hand side of the four-lane kernels go (no GPU needed): builds the quad module of a
system with -DHAMK_PROBE_PHASE (a numbered s_setprio at every phase boundary of hamk_quad.hpp), disassembles hamk_hameqs_k and
prints the instruction classes between consecutive markers.
  python scripts/quad_phases.py chain32"""
import collections
import os
os.environ["HAMK_TEST_OVERRIDES"] = "1"               # this script drives libhamk.so through its HAMK_* test overrides
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
d = tempfile.mkdtemp(); os.chmod(d, 0o700); os.environ["HAMK_CACHE_DIR"] = d
os.environ["HAMK_HIPRTC_FLAGS"] = (os.environ.get("HAMK_HIPRTC_FLAGS", "") + " -DHAMK_PROBE_NO_SLOWPATH -DHAMK_PROBE_PHASE").strip()
from hamilton_amd import _abi, api, examples
import isa_stats

name = sys.argv[1] if len(sys.argv) > 1 else "chain32"
s = api.system_from_spec(examples.get(name), {"mapping": _abi.MAP_QUAD})
info = {l.split()[0]: l for l in s.build_info.splitlines()}
kern = sys.argv[2] if len(sys.argv) > 2 else "hamk_hameqs_k"
ins = isa_stats.disassemble(s.code_object(1 if "no-machine-licm" in info[kern] else 0))[kern]
cuts = [i for i, (_, mn, _) in enumerate(ins) if mn == "s_setprio"]
bounds = [0] + cuts + [len(ins)]
print(f"{name}: {len(ins)} instructions in {kern}, {len(cuts)} phase markers")
for a, b in zip(bounds, bounds[1:]):
    h = collections.Counter(isa_stats.classify(mn) for _, mn, _ in ins[a:b])
    dpp = sum(1 for _, mn, ops in ins[a:b] if "dpp" in mn or "quad_perm" in ops)
    acc = sum(1 for _, mn, _ in ins[a:b] if mn.startswith("v_accvgpr"))
    if b - a > 20:
        print(f"  [{a:6d},{b:6d}) n={b - a:5d} f64={h['valu_f64']:5d} mov={h['valu_mov']:5d} (dpp {dpp}, accvgpr {acc}) sel={h['valu_sel']:4d} lds={h.get('lds', 0):4d} scratch={h.get('scratch', 0):3d} other={h['valu_other'] + h['valu_cmp']:4d}")
