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


def hottest_loop(ins):
    """Longest span closed by a backward conditional/unconditional branch."""
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    best = None
    for i, (a, mn, ops) in enumerate(ins):
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
