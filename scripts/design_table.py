#!/usr/bin/env python3
"""Prints the rows of DESIGN.md section 4's table from profiles/<tag>_bench_configs.jsonl (one bench line per BASELINE config).
  python scripts/design_table.py [r06]"""
import json
import math
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
SUP = str.maketrans("0123456789-", "⁰¹²³⁴⁵⁶⁷⁸⁹⁻")


def sci(x, digits=2):
    e = int(math.floor(math.log10(x)))
    m = x / 10 ** e
    return f"{m:.{digits}f}·10{str(e).translate(SUP)}"


rows = {}
for l in open(os.path.join(ROOT, "profiles", f"{tag}_bench_configs.jsonl")):
    d = json.loads(l)
    rows[d["config"]["workload"].split()[0]] = d
for cfg, key, label, B in (("C2", "doublePendulum", "doublePendulum (4,2)", "2²⁰"), ("C3", "twoBody", "twoBody (4,2)", "2²⁰"), ("C3", "spring", "spring (3,3)", "2²⁰"),
                           ("C4", "threeBodyPolar", "threeBodyPolar (6,6)", "2¹⁸"), ("C5", "chain8", "chain8 (16,8), lane", "2¹⁶"),
                           ("C5", "chain16", "chain16 (32,16), lane", "2¹⁶"), ("C5", "chain32", "chain32 (64,32), quad", "2¹⁶")):
    d = rows[key]; r = d["roofline"]; fp = r["fp64"]; cb = d["cpu_baseline"]; one = cb.get("single_thread", {}).get("value")
    flagged = d.get("status_flagged")
    valu = f"{fp['valu_insts_per_wave_step']:,}".replace(",", " ")
    print(f"| {cfg} | {label} | {B} | {sci(d['value'])} | {r['frac']:.3f} | {fp['achieved_tflops']:.1f} ({fp['frac_of_peak']:.2f}) | {valu}"
          f" | {fp['valu_issue_frac']:.2f} / {fp['valu_issue_frac_at_measured_clock']:.2f} | {sci(cb['value'], 1)} / {sci(one, 1) if one else '—'} | "
          f"{100.0 * flagged / d['config']['trajectories_per_gpu']:.0f} % |")
