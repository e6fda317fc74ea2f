#!/usr/bin/env python3
"""Throughput of the RK4 kernel against the ensemble size B, for every mapping a system supports.

  python scripts/sweep_batch.py [--out profiles/r03_throughput_vs_B.jsonl] [--systems chain8,chain16,...]
  python scripts/sweep_batch.py --predict profiles/r03_throughput_vs_B.jsonl        (no GPU: prints the tables)

Why: BASELINE.json's configs 3 and 4 shard a FIXED ensemble over 1 -> 8 GPUs (GPU g of G owns
[g B/G, (g+1) B/G)), so at G = 8 a GPU holds 32 768 (C4) or 8 192 (C5) trajectories -- 512 / 128 wavefronts of the
one-trajectory-per-lane kernels for 1024 SIMDs.  This measures steps/s at B = 2^13 ... 2^20 on the lane and the
wave-cooperative kernels (hamk_options::mapping), from which (a) the library's per-launch choice of the mapping
(hamk_dispatch.cpp choose_mapping) and (b) the predicted strong-scaling curve of each fixed-size config follow.
"""
from __future__ import annotations

import argparse
import json
import os
os.environ["HAMK_TEST_OVERRIDES"] = "1"               # this script drives libhamk.so through its HAMK_* test overrides
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIG_B = {"doublePendulum": 1 << 20, "twoBody": 1 << 20, "spring": 1 << 20, "threeBodyPolar": 1 << 18,
            "chain8": 1 << 16, "chain16": 1 << 16, "chain32": 1 << 16}


def measure(args):
    import torch
    from hamilton_amd import _abi, api, examples
    names = args.systems.split(",")
    out = open(args.out, "a")
    for name in names:
        spec = examples.get(name)
        maps = [("lane", _abi.MAP_LANE)] if spec.n <= 16 else []
        if spec.n <= 32 and spec.n >= 4 and "quad" in args.mappings:
            maps.append(("quad", _abi.MAP_QUAD))
        if "wave" in args.mappings:
            maps.append(("wave", _abi.MAP_WAVE))
        maps = [m for m in maps if m[0] in args.mappings]
        top = max(CONFIG_B.get(name, 1 << 16), 1 << 16)
        for label, mp in maps:
            if label == "wave" and spec.n <= 3:
                continue                                        # never competitive: 16 lanes for a 2 x 2 solve
            s = api.system_from_spec(spec, {"mapping": mp})
            B = 1 << 13
            while B <= top:
                q, qd = examples.sample_config(spec, 0, B)
                ph = api.toPhase(s, api.Config(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()))
                st = api.Phase(ph.positions.clone(), ph.momenta.clone())
                nsteps = 4
                api.rk4Steps(spec.dt, nsteps, s, st, inplace=True)        # load + self-check + warm
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                api.rk4Steps(spec.dt, nsteps, s, st, inplace=True)
                torch.cuda.synchronize()
                probe = (time.perf_counter() - t0) / nsteps
                nsteps = int(max(4, min(2000, args.launch_ms * 1e-3 / max(probe, 1e-9))))
                best = None
                for _ in range(args.repeats):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    api.rk4Steps(spec.dt, nsteps, s, st, inplace=True)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1)
                    best = ms if best is None else min(best, ms)
                rec = {"system": name, "n": spec.n, "mapping": label, "B": B, "rk4_steps_per_launch": nsteps,
                       "kernel_ms": best, "steps_per_s": B * nsteps / (best * 1e-3)}
                print(json.dumps(rec), flush=True)
                out.write(json.dumps(rec) + "\n")
                out.flush()
                B *= 2
    out.close()


def predict(path):
    recs = [json.loads(l) for l in open(path) if l.strip()]
    by = {}
    for r in recs:
        by.setdefault(r["system"], {}).setdefault(r["mapping"], {})[r["B"]] = r["steps_per_s"]
    print("best mapping by ensemble size (steps/s):")
    for name, maps in by.items():
        Bs = sorted({b for m in maps.values() for b in m})
        row = []
        for b in Bs:
            cand = {k: v[b] for k, v in maps.items() if b in v}
            k = max(cand, key=cand.get)
            row.append(f"{b}:{k}:{cand[k]:.3g}")
        print(f"  {name:16s} " + "  ".join(row))
    print("\npredicted strong scaling of the fixed-size configs (per-GPU B = B_config / G, best mapping per launch; the")
    print("shards are independent, so the aggregate is G x the per-GPU rate at that B):")
    for name, maps in by.items():
        Bc = CONFIG_B.get(name)
        if not Bc:
            continue
        base = None
        for G in (1, 2, 4, 8):
            b = Bc // G
            cand = {k: v[b] for k, v in maps.items() if b in v}
            if not cand:
                continue
            k = max(cand, key=cand.get)
            agg = G * cand[k]
            base = base or agg
            print(f"  {name:16s} G={G}  B/G={b:8d}  {k:5s} per-GPU {cand[k]:.3g}  aggregate {agg:.3g}  efficiency {agg / (G * base):.2f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_throughput_vs_B.jsonl"))
    ap.add_argument("--systems", default="threeBodyPolar,chain8,chain16,chain32,spring")
    ap.add_argument("--launch-ms", type=float, default=40.0)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--mappings", default="lane,quad,wave")
    ap.add_argument("--predict", default=None)
    a = ap.parse_args()
    if a.predict:
        predict(a.predict)
    else:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        measure(a)
