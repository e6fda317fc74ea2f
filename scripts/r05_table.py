#!/usr/bin/env python3
"""Markdown rows for DESIGN.md section 4 from the final pass's records (profiles/r05_bench_configs.jsonl, r05_bench_stepham.jsonl,
r05_bench_wave.jsonl and the r05_*_summary.json of scripts/summarize_profile.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "profiles")


def lines(name):
    path = os.path.join(P, name)
    return [json.loads(l) for l in open(path) if l.strip().startswith("{")] if os.path.exists(path) else []


def summary(system, suffix=""):
    path = os.path.join(ROOT, "profiles", f"r05_{system}{suffix}_summary.json")
    return json.load(open(path)) if os.path.exists(path) else {}


print("| system (m,n) | B | RK4 steps/s | §8d fraction | fp64 of 78.6 TF | VALU / wave-step | issue @2.4 GHz / @measured clock (MHz) | cpu_baseline all cores / one | parity 1 step / 100 steps all lanes | flagged |")
print("|---|---|---|---|---|---|---|---|---|---|")
for d in lines("r05_bench_configs.jsonl") + lines("r05_bench_wave.jsonl"):
    r, f = d["roofline"], d["roofline"]["fp64"]
    cb, par = d.get("cpu_baseline", {}), d.get("parity", {})
    k100 = [k for k in par if k.startswith("max_abs_dphase_100_steps_all")]
    print(f"| {d['config']['workload'].split(' ensemble')[0]} | {d['config']['trajectories_per_gpu']} | {d['value']:.3g} | {r['frac']:.3f} | "
          f"{f.get('achieved_tflops', 0):.1f} ({f.get('frac_of_peak', 0):.2f}) | {f.get('valu_insts_per_wave_step')} | "
          f"{f.get('valu_issue_frac', 0):.2f} / {(f.get('valu_issue_frac_at_measured_clock') or 0):.2f} ({f.get('sclk_mhz_during_timed_region')}) | "
          f"{cb.get('value', 0):.3g} / {cb.get('single_thread', {}).get('value', 0):.3g} | {par.get('max_abs_dphase_1_step', 0):.1e} / {(par.get(k100[0]) if k100 else 0) or 0:.1e} | "
          f"{d.get('status_flagged_drift', 0) / max(1, d['config']['trajectories_per_gpu']):.0%} |")
print()
print("| system | stepHam/s | mean / wave-max sub-steps | lane utilisation | identical sub-step counts vs oracle | max |dphase| |")
print("|---|---|---|---|---|---|")
for d in lines("r05_bench_stepham.jsonl"):
    dv, par = d.get("divergence", {}), d.get("parity", {})
    print(f"| {d['config']['workload'].split(' ensemble')[0]} (B {d['config']['trajectories_per_gpu']}) | {d['value']:.3g} | {dv.get('mean_substeps_per_lane', 0):.2f} / {dv.get('mean_of_wave_max_substeps', 0):.2f} | "
          f"{dv.get('lane_utilisation', 0):.2f} | {par.get('identical_substep_counts_frac')} | {par.get('max_abs_dphase_lanes_with_identical_counts')} |")
print()
for system, suffix in [("doublePendulum", ""), ("twoBody", ""), ("spring", ""), ("threeBodyPolar", ""), ("chain8", ""), ("chain16", ""), ("chain32", ""),
                       ("dense32", ""), ("chain64", ""), ("chain8", "_stepham"), ("chain16", "_stepham")]:
    s = summary(system, suffix)
    if not s:
        continue
    keys = ("valu_pipe_busy_frac_of_1024_simds", "sq_wait_any_frac_of_wave_cycles", "sq_active_inst_any_frac_of_wave_cycles", "wait_frac_of_wave_cycles",
            "lds_busy_frac_of_256_units", "valu_insts_per_wave_per_rk4_step")
    hbm = s.get("hbm", {})
    print(system + suffix, s.get("rocprofv3_kernel_stats", {}).get("AverageNs"), "ns;", {k: round(s[k], 3) for k in keys if k in s},
          "scratch", s.get("dispatch", {}).get("Scratch_Size"), "HBM/launch %.4g MB (x%.2f of state)" % (hbm.get("hbm_bytes_per_launch", 0) / 1e6,
          hbm.get("hbm_bytes_per_launch", 0) / max(1, hbm.get("expected_read_bytes", 1) + hbm.get("expected_write_bytes", 1))) if hbm else "")
