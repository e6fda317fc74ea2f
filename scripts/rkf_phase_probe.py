#!/usr/bin/env python3
"""Where do the cycles of the parked adaptive stepper go?  (round 5, VERDICT r4 item 1: measure before code.)

For each system, one `stepHam dt` launch over the bench ensemble (device sampler, the bench's own initial conditions):
  * the shipped kernel: launch time, sub-step statistics, lane utilisation of the attempt loop, at several ensemble sizes
    (does the rate scale with the number of busy CUs, or is something chip-wide shared?);
  * candidate builds (-D switches through HAMK_HIPRTC_FLAGS), same launch, states compared with the shipped kernel's;
  * two probe builds (-DHAMK_PROBE_CYC=1|2, hamk_device.hpp): s_memtime at the phase boundaries of an attempt, summed per lane
    and returned through nsub / status: cycles of the stage combinations (incl. waiting for their row loads), of the
    right-hand sides, of the stores of a stage's result, of the controller + commit.
Everything is compiled ahead (`--compile-only`, no GPU) into .hamk_cache/ so that the GPU box only measures.
  python scripts/rkf_phase_probe.py [--compile-only] [--systems chain8,chain16] > gpurun_out/r05_rkf_phase_probe.jsonl"""
import json
import os
os.environ["HAMK_TEST_OVERRIDES"] = "1"               # this script drives libhamk.so through its HAMK_* test overrides
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
COMPILE_ONLY = "--compile-only" in sys.argv
os.environ.setdefault("HAMK_CACHE_DIR", os.path.join(ROOT, ".hamk_cache"))
os.environ["HAMK_SELFCHECK"] = "0"                          # probe builds return cycle counts where the self-check expects a status
from hamilton_amd import _abi, api, examples                # noqa: E402

SYSTEMS = ["chain8", "chain16"]
for a in sys.argv[1:]:
    if a.startswith("--systems="):
        SYSTEMS = a.split("=", 1)[1].split(",")
CANDIDATES = [("shipped", ""), ("pair-rows", "-DHAMK_PAIR_ROWS=1")]
for a in sys.argv[1:]:
    if a.startswith("--candidate="):                        # --candidate=tag:flags
        tag, fl = a.split("=", 1)[1].split(":", 1)
        CANDIDATES.append((tag, fl.replace("_-D", " -D")))      # "_-D" stands for " -D" (shell quoting)
PROBES = [] if "--no-probes" in sys.argv else [("cyc1", "-DHAMK_PROBE_CYC=1"), ("cyc2", "-DHAMK_PROBE_CYC=2")]
SIZES = [65536] if "--one-size" in sys.argv else [16384, 32768, 65536, 131072]
if not COMPILE_ONLY:
    import torch


def build(spec, flags):
    os.environ["HAMK_HIPRTC_FLAGS"] = flags
    return api.system_from_spec(spec, {"mapping": _abi.MAP_LANE})


def state(s, spec, B):
    cfg = api.sampleConfig(s, spec.q_box, spec.qd_box, 0, B, examples.SEED, torch.device("cuda", 0))
    ph = api.toPhase(s, api.Config(cfg.positions, cfg.velocities))
    # a few calls in: the bench times steady state, not the first call from rest
    for _ in range(3):
        ph = api.stepHam(spec.dt, s, ph)
    return ph


def timed(s, spec, ph, reps=6):
    out = api.stepHam(spec.dt, s, ph)
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = api.stepHam(spec.dt, s, ph); e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return min(ms), sorted(ms)[len(ms) // 2], out


for name in SYSTEMS:
    spec = examples.get(name)
    built = {tag: build(spec, fl) for tag, fl in CANDIDATES + PROBES}
    if COMPILE_ONLY:
        for tag, s in built.items():
            print(name, tag, [l for l in s.build_info.splitlines() if l.startswith("hamk_rkf45_k")], flush=True)
        continue
    ref = built["shipped"]
    for B in SIZES:
        ph = state(ref, spec, B)
        best, med, out0 = timed(ref, spec, ph)
        nsub = ref.last_nsub.double()
        wmax = nsub[: (B // 64) * 64].reshape(-1, 64).amax(1)
        rec = {"what": "shipped", "system": name, "B": B, "ms_best": best, "ms_median": med, "calls_per_s": B / (best * 1e-3),
               "mean_substeps": float(nsub.mean()), "mean_wave_max_substeps": float(wmax.mean()), "lane_utilisation": float(nsub.mean() / wmax.mean()),
               "wave_attempt_us": best * 1e3 / float(wmax.mean())}
        print(json.dumps(rec), flush=True)
        if B != 65536:
            continue
        # the same right-hand side inside the fixed-step kernel (4 per step, nothing parked in scratch): its time per wavefront
        api.rk4Steps(spec.dt, 5, ref, ph)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); api.rk4Steps(spec.dt, 50, ref, ph); e1.record()
        torch.cuda.synchronize()
        print(json.dumps({"what": "rk4_reference", "system": name, "B": B, "rk4_rhs_us_per_wavefront": e0.elapsed_time(e1) * 1e3 / 200.0,
                          "stepham_attempt_us_per_wavefront": rec["wave_attempt_us"], "stepham_us_per_rhs": rec["wave_attempt_us"] / 6.0}), flush=True)
        for tag, fl in CANDIDATES[1:]:
            s = built[tag]
            b2, m2, out = timed(s, spec, ph)
            d = float(max((out.positions - out0.positions).abs().max(), (out.momenta - out0.momenta).abs().max()))
            same = bool(torch.equal(s.last_nsub, ref.last_nsub))
            print(json.dumps({"what": "candidate", "system": name, "B": B, "variant": tag, "flags": fl, "ms_best": b2, "ms_median": m2,
                              "calls_per_s": B / (b2 * 1e-3), "vs_shipped": best / b2, "max_abs_diff_to_shipped": d, "same_substeps": same}), flush=True)
        if not PROBES:
            continue
        cyc = {}
        for tag, fl in PROBES:
            s = built[tag]
            b2, m2, _ = timed(s, spec, ph, reps=3)
            a = s.last_nsub.double()[: (B // 64) * 64].reshape(-1, 64).amax(1)       # the lane that ran every attempt of its wavefront
            b = s.last_status.double()[: (B // 64) * 64].reshape(-1, 64).amax(1)
            cyc[tag] = (float(a.mean()), float(b.mean()), b2)
        att = float(wmax.mean())
        combo, rhs_c = cyc["cyc1"][0], cyc["cyc1"][1]
        put, tail = cyc["cyc2"][0], cyc["cyc2"][1]
        tot = combo + rhs_c + put + tail
        print(json.dumps({"what": "phase_cycles", "system": name, "B": B, "wave_attempts": att,
                          "probe_ms": [cyc["cyc1"][2], cyc["cyc2"][2]], "shipped_ms": best,
                          "cycles_per_launch": {"combination_and_row_loads": combo, "right_hand_sides": rhs_c, "row_stores": put, "control_commit_entry_exit": tail,
                                                "sum": tot},
                          "fraction": {"combination_and_row_loads": combo / tot, "right_hand_sides": rhs_c / tot, "row_stores": put / tot,
                                       "control_commit_entry_exit": tail / tot},
                          "cycles_per_rhs": rhs_c / (6 * att), "cycles_per_stage_combination": combo / (6 * att),
                          "implied_clock_ghz": tot / (cyc["cyc1"][2] * 1e-3) / 1e9}), flush=True)
