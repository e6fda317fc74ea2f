#!/usr/bin/env python3
"""Static ISA statistics of a system's gfx950 kernels (works without a GPU).

  python scripts/isa_stats.py doublePendulum [kernel-substring]

JIT-compiles the system into a private cache directory, disassembles the code object with
llvm-objdump and prints, per kernel: code bytes, registers/scratch from the kernel descriptor
notes, and an instruction-class histogram of (a) the whole kernel and (b) its hottest loop
(the longest backward-branch span), which for hamk_rk4_steps_k is one RK4 step (unrolled
body) or one stage (stage-loop body).
"""
from __future__ import annotations

import collections
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def classify(mn: str) -> str:
    if mn.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64",
                      "v_div_", "v_ldexp_f64", "v_frexp", "v_trig_preop", "v_fract_f64", "v_rndne_f64", "v_floor_f64",
                      "v_ceil_f64", "v_trunc_f64", "v_max_f64", "v_min_f64", "v_cvt_f64", "v_cvt_i32_f64", "v_cvt_u32_f64")):
        return "valu_f64"
    if mn.startswith("v_cmp") or mn.startswith("v_cmpx"):
        return "valu_cmp"
    if mn.startswith(("v_mov", "v_accvgpr", "v_swap")):
        return "valu_mov"
    if mn.startswith(("v_cndmask",)):
        return "valu_sel"
    if mn.startswith("v_"):
        return "valu_other"
    if mn.startswith("s_waitcnt") or mn.startswith("s_nop"):
        return "wait/nop"
    if mn.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if mn.startswith("scratch_"):
        return "scratch"
    return "other"


def parse(disasm: str):
    kernels = {}
    cur = None
    for line in disasm.splitlines():
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
        if m:
            cur = m.group(2)
            kernels[cur] = []
            continue
        m = re.match(r"^\s+(\S+)\s+(.*?)//\s*([0-9A-Fa-f]+):(.*)$", line)
        if m and cur is not None:
            kernels[cur].append((int(m.group(3), 16), m.group(1), m.group(2).strip() + " " + m.group(4)))
    return kernels


def long_branch_target(ins, i):
    """Target of the long-branch sequence s_getpc_b64 / s_add_u32 lo, lo, literal / s_addc_u32 hi, hi, literal / s_setpc_b64 that ends at
    ins[i] (the back edge of a loop whose body exceeds the +-128 KiB of s_branch: the stage loops of the dense maps), or None."""
    if ins[i][1] != "s_setpc_b64" or i < 3 or ins[i - 3][1] != "s_getpc_b64" or ins[i - 2][1] != "s_add_u32" or ins[i - 1][1] != "s_addc_u32":
        return None
    lit = lambda ops: ops.split()[2].rstrip(",") if len(ops.split()) > 2 else None
    try:
        lo = int(lit(ins[i - 2][2]), 0) & 0xFFFFFFFF
        hi = int(lit(ins[i - 1][2]), 0) & 0xFFFFFFFF
    except (TypeError, ValueError):
        return None
    off = (hi << 32) | lo
    if off >= 1 << 63:
        off -= 1 << 64
    return ins[i - 3][0] + 4 + off


def hottest_loop(ins):
    """Longest span closed by a backward conditional/unconditional branch (short form or the long-branch sequence)."""
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    best = None
    for i, (a, mn, ops) in enumerate(ins):
        if mn == "s_setpc_b64":
            tgt = long_branch_target(ins, i)
            if tgt is not None and tgt in addr_index and tgt <= a:
                j = addr_index[tgt]
                if best is None or (i - j) > (best[1] - best[0]):
                    best = (j, i)
            continue
        if not mn.startswith(("s_cbranch", "s_branch")):
            continue
        tgt = None
        m2 = re.search(r"\+0x([0-9a-f]+)>", ops)
        if m2:
            tgt = ins[0][0] + int(m2.group(1), 16)
        if tgt is None or tgt not in addr_index or tgt > a:
            continue
        j = addr_index[tgt]
        if best is None or (i - j) > (best[1] - best[0]):
            best = (j, i)
    return best


FP64_FLOPS = {"v_fma_f64": 2, "v_fmac_f64": 2, "v_mul_f64": 1, "v_add_f64": 1, "v_rcp_f64": 1, "v_rsq_f64": 1, "v_sqrt_f64": 1,
              "v_div_fmas_f64": 2, "v_div_fixup_f64": 1, "v_div_scale_f64": 1, "v_ldexp_f64": 1, "v_max_f64": 1, "v_min_f64": 1,
              "v_fract_f64": 1, "v_rndne_f64": 1, "v_floor_f64": 1, "v_ceil_f64": 1, "v_trunc_f64": 1, "v_trig_preop_f64": 1}
MFMA_F64_FLOPS_PER_LANE = {"v_mfma_f64_16x16x4_f64": 16 * 16 * 4 * 2 // 64, "v_mfma_f64_4x4x4_4b_f64": 4 * 4 * 4 * 4 * 2 // 64}


def guarded_mask(ins, span):
    """Instructions of ins[span] that sit behind a forward conditional branch inside the span (the body
    of an `if`: rare library / fallback paths, or wave-uniform stage selection): True = guarded."""
    lo, hi = span
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    mask = [False] * (hi - lo + 1)
    for i in range(lo, hi + 1):
        a, mn, ops = ins[i]
        if not mn.startswith("s_cbranch"):
            continue
        m2 = re.search(r"\+0x([0-9a-f]+)>", ops)
        if not m2:
            continue
        tgt = ins[0][0] + int(m2.group(1), 16)
        if tgt in addr_index and i < addr_index[tgt] <= hi + 1:
            for k in range(i + 1, addr_index[tgt]):
                mask[k - lo] = True
    return mask


def loop_stats(ins, span=None):
    """Histogram of one span (default: the hottest loop) with the fp64 flop count of its straight-line
    (unguarded) part: 2 per FMA, 1 per mul/add/rcp/..., 32 per lane per v_mfma_f64_16x16x4."""
    span = span or hottest_loop(ins)
    if span is None:
        return None
    mask = guarded_mask(ins, span)
    h_all, h_main = collections.Counter(), collections.Counter()
    flops_main = flops_all = 0
    for k, (_, mn, _) in enumerate(ins[span[0]:span[1] + 1]):
        base = mn[:-4] if mn.endswith(("_e32", "_e64")) else mn
        base = base[:-5] if base.endswith(("_dpp", "_sdwa")) and False else base
        f = FP64_FLOPS.get(base, 0) + MFMA_F64_FLOPS_PER_LANE.get(base, 0)
        c = classify(mn)
        h_all[c] += 1
        flops_all += f
        if not mask[k]:
            h_main[c] += 1
            flops_main += f
    valu = lambda h: sum(v for c, v in h.items() if c.startswith("valu"))
    return {"instructions": sum(h_all.values()), "valu": valu(h_all), "valu_f64": h_all["valu_f64"], "fp64_flops": flops_all,
            "unguarded": {"instructions": sum(h_main.values()), "valu": valu(h_main), "valu_f64": h_main["valu_f64"],
                          "fp64_flops": flops_main, "lds": h_main["lds"], "salu": h_main["salu"], "scratch": h_main["scratch"],
                          "mfma": sum(1 for k, (_, mn, _) in enumerate(ins[span[0]:span[1] + 1]) if not mask[k] and mn.startswith("v_mfma"))},
            "histogram": dict(h_all)}


def disassemble(code_object: bytes):
    """{kernel name: [(address, mnemonic, operands), ...]} of a gfx950 ELF."""
    with tempfile.NamedTemporaryFile(suffix=".hsaco") as fh:
        fh.write(code_object)
        fh.flush()
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", fh.name], capture_output=True, text=True).stdout
    return parse(dis)


def rk4_step_stats(spec, system=None):
    """Cost of ONE RK4 step of one wavefront for the system `spec` (hamilton_amd.examples.SystemSpec),
    counted from code objects of the same source as `system` (the module in use: same AD mode, same
    stepping body, same sincos chain length K), built with probe defines that make the count a
    count of what a lane of a healthy ensemble EXECUTES:
      -DHAMK_PROBE_NO_SLOWPATH  removes the bodies of the rare branches (library sin/cos for
                                |x| >= 1.6e6 or NaN, far-from-anchor re-evaluation, ...: ~2000
                                instructions inside the stepping loop that are never executed);
      -DHAMK_PROBE_TRIG=m       (stage-loop bodies only) fixes the wave-uniform sincos mode of the
                                loop's one evaluation at compile time (m = 0 full anchor, 1 narrow
                                rotation + new anchor, 2 narrow, 3 short), so the sides of that scalar
                                switch are not all counted.
    The hottest loop is one step (unrolled body: the four stage modes are compile-time constants
    there) or one stage (stage-loop / wave bodies).  A step costs
      unrolled    L                                  L: the loop
      stage loop  L(0) + L(1) + L(2) + L(3)          L(m): the loop with the sincos mode fixed to m
    Cross-check: PMC SQ_INSTS_VALU per wave per step (profiles/*_summary.json).
    Returns None when llvm-objdump is unavailable."""
    if not os.path.exists(OBJDUMP):
        return None
    from hamilton_amd import api
    if system is None:
        system = api.system_from_spec(spec)
    src = system.source
    wave = "HAMK_INSTANTIATE_WAVE" in src
    quad = "HAMK_INSTANTIATE_QUAD" in src
    stage_loop = "RK4_STAGE_LOOP = true" in src or wave or quad
    m = re.search(r"NTRIG_F = (\d+)", src)
    chained = (not wave) and (not quad) and m is not None and 1 <= int(m.group(1)) <= 4          # hamk_device.hpp StageTrig
    base_env = {"HAMK_RK4_LOOP": "1" if "RK4_STAGE_LOOP = true" in src else "0", "HAMK_WAVE": "1" if wave else "0", "HAMK_QUAD": "1" if quad else "0",
                "HAMK_AD_MODE": "H" if "MODE_H = true" in src else ("R" if "MODE_R = true" in src else "D")}

    def count(trig):
        flags = (os.environ.get("HAMK_HIPRTC_FLAGS", "") + " -DHAMK_PROBE_NO_SLOWPATH").strip()
        if trig is not None:
            flags += f" -DHAMK_PROBE_TRIG={trig}"
        env = dict(base_env, HAMK_HIPRTC_FLAGS=flags, HAMK_TEST_OVERRIDES="1")
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            probe = api.system_from_spec(spec)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        info = {l.split()[0]: l.split()[1] for l in probe.build_info.splitlines() if l}
        co = probe.code_object(1 if "no-machine-licm" in info.get("hamk_rk4_steps_k", "") else 0)
        ins = disassemble(co).get("hamk_rk4_steps_k") if co else None
        st = loop_stats(ins) if ins else None
        if st is None:
            return None
        lo, hi = hottest_loop(ins)
        return {"valu": st["valu"], "valu_f64": st["valu_f64"], "flops": st["fp64_flops"],
                "mfma": sum(1 for _, mn, _ in ins[lo:hi + 1] if mn.startswith("v_mfma")),
                "lds": st["histogram"].get("lds", 0), "scratch": st["histogram"].get("scratch", 0)}

    if chained and stage_loop:
        L = [count(m) for m in (0, 1, 2, 3)]
        if any(x is None for x in L):
            return None
        w = {k: sum(x[k] for x in L) for k in L[0]}
        detail = {"loop_full_anchor": L[0], "loop_narrow_anchor": L[1], "loop_narrow": L[2], "loop_short": L[3]}
    else:
        A = count(None)
        if A is None:
            return None
        w = {k: A[k] * (4 if stage_loop else 1) for k in A}
        detail = {"loop": A}
    return {"loop_is": "one stage (x4 per step)" if stage_loop else "one RK4 step",
            "valu_per_wave_step": w["valu"], "valu_f64_per_wave_step": w["valu_f64"], "fp64_flops_per_lane_step": w["flops"],
            "mfma_per_wave_step": w["mfma"], "lds_per_wave_step": w["lds"], "scratch_per_wave_step": w["scratch"], "detail": detail,
            "source": "llvm-objdump of this system's module built with -DHAMK_PROBE_NO_SLOWPATH (rare library/fallback branch bodies "
                      "removed) and -DHAMK_PROBE_TRIG (wave-uniform sincos case fixed), stepping loop of hamk_rk4_steps_k, weighted per step"}


def rkf45_attempt_stats(spec, system=None):
    """Cost of ONE attempt of the adaptive stepper (hamk_rkf45_k; six right-hand sides + controller) of one wavefront,
    counted from a probe build of the same module: -DHAMK_PROBE_NO_SLOWPATH removes the never-executed library branches,
    -DHAMK_PROBE_MARK brackets the attempt (s_setprio 1 ... 2) and, in the stage-loop body, the one inlined right-hand
    side (s_setprio 3 ... 0), which an attempt executes six times.  Lane kernels, and the quad kernels' parked body (a wavefront
    then holds 16 trajectories)."""
    if not os.path.exists(OBJDUMP):
        return None
    from hamilton_amd import api
    if system is None:
        system = api.system_from_spec(spec)
    src = system.source
    if "HAMK_INSTANTIATE_WAVE" in src:
        return None
    quad = "HAMK_INSTANTIATE_QUAD" in src
    env = {"HAMK_RKF_LOOP": "1" if ("RKF_STAGE_LOOP = true" in src or quad) else "0", "HAMK_WAVE": "0", "HAMK_QUAD": "1" if quad else "0",
           "HAMK_RKF_PARK": "1" if ("HAMK_RKF_PARK 1" in src or "HAMK_QUAD_RKF_PARK 1" in src) else "0",
           "HAMK_AD_MODE": "H" if "MODE_H = true" in src else ("R" if "MODE_R = true" in src else "D"),
           "HAMK_TEST_OVERRIDES": "1", "HAMK_HIPRTC_FLAGS": (os.environ.get("HAMK_HIPRTC_FLAGS", "") + " -DHAMK_PROBE_NO_SLOWPATH -DHAMK_PROBE_MARK").strip()}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        probe = api.system_from_spec(spec)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    info = {l.split()[0]: l.split()[1] for l in probe.build_info.splitlines() if l}
    co = probe.code_object(1 if "no-machine-licm" in info.get("hamk_rkf45_k", "") else 0)
    ins = disassemble(co).get("hamk_rkf45_k") if co else None
    if not ins:
        return None
    where = {}
    for i, (_, mn, ops) in enumerate(ins):
        if mn == "s_setprio":
            where.setdefault(int(ops.split()[0]), []).append(i)
    if 1 not in where or 2 not in where:
        return None
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    back = []                                               # (head, branch) of every backward branch
    for i, (a, mn, ops) in enumerate(ins):
        if mn.startswith(("s_cbranch", "s_branch")):
            m2 = re.search(r"\+0x([0-9a-f]+)>", ops)
            tgt = addr_index.get(ins[0][0] + int(m2.group(1), 16)) if m2 else None
            if tgt is not None and tgt <= i:
                back.append((tgt, i))

    def region(begin, end):
        """Indices executed from marker `begin` to marker `end`.  When the loop that holds them is laid out rotated (its
        last marker first) the path runs on to the loop's back edge and continues from the loop's head."""
        if begin <= end:
            return set(range(begin, end + 1))
        holds = [(hi - lo, lo, hi) for lo, hi in back if lo <= end and hi >= begin]
        if not holds:
            return set()
        _, lo, hi = min(holds)
        return set(range(begin, hi + 1)) | set(range(lo, end + 1))

    body = region(min(where[1]), max(where[2]))
    rhs_idx = region(min(where[3]), max(where[0])) if (3 in where and 0 in where) else set()
    once = collections.Counter(classify(ins[i][1]) for i in body if i not in rhs_idx)
    rhs = collections.Counter(classify(ins[i][1]) for i in rhs_idx)
    rhs_spans = bool(rhs_idx)
    tot = collections.Counter()
    for k, v in once.items():
        tot[k] += v
    for k, v in rhs.items():
        tot[k] += 6 * v
    valu = sum(v for c, v in tot.items() if c.startswith("valu"))
    return {"valu_per_wave_attempt": valu, "valu_f64_per_wave_attempt": tot.get("valu_f64", 0), "lds_per_wave_attempt": tot.get("lds", 0),
            "scratch_per_wave_attempt": tot.get("scratch", 0), "body": "stage loop (one inlined right-hand side x 6)" if rhs_spans else "unrolled",
            "source": "llvm-objdump of hamk_rkf45_k built with -DHAMK_PROBE_NO_SLOWPATH -DHAMK_PROBE_MARK: one attempt between its markers"}


def main():
    from hamilton_amd import api, examples
    name = sys.argv[1] if len(sys.argv) > 1 else "doublePendulum"
    want = sys.argv[2] if len(sys.argv) > 2 else "rk4"
    with tempfile.TemporaryDirectory() as d:
        os.environ["HAMK_CACHE_DIR"] = d
        os.environ.pop("HAMK_CACHE", None)
        s = api.system_from_spec(examples.get(name))
        files = glob.glob(os.path.join(d, "*.hsaco"))
        assert files, "no code object produced"
        co = max(files, key=os.path.getmtime)
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True).stdout
    kernels = parse(dis)
    meta = {}
    for blk in notes.split(".name:")[1:]:
        nm = blk.split()[0]
        g = lambda k: (re.search(rf"\.{k}:\s*(\d+)", blk) or [None, "?"])[1]
        meta[nm] = dict(vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"),
                        scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"),
                        sgpr_spill=g("sgpr_spill_count"), vgpr_spill=g("vgpr_spill_count"))
    for k, ins in kernels.items():
        if want not in k:
            continue
        print(f"== {k}: {len(ins)} instructions, {ins[-1][0] - ins[0][0] + 4} bytes, {meta.get(k, {})}")
        for label, span in (("kernel", (0, len(ins) - 1)), ("hottest loop", hottest_loop(ins))):
            if span is None:
                print("   no loop found")
                continue
            h = collections.Counter(classify(mn) for _, mn, _ in ins[span[0]:span[1] + 1])
            tot = sum(h.values())
            valu = sum(v for c, v in h.items() if c.startswith("valu"))
            print(f"   {label:13s} n={tot:6d} VALU={valu:6d}  " + "  ".join(f"{c}={v}" for c, v in sorted(h.items())))
    del s


if __name__ == "__main__":
    main()
