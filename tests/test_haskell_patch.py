"""The reference-side patch of the Haskell shim (bindings/haskell/hamilton-hip.patch) applies cleanly to
the reference tree -- where that tree exists (this container; the GPU box does not have it: skipped)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present")
def test_patch_applies_to_the_reference_tree(tmp_path):
    work = tmp_path / "ref"
    os.makedirs(work)
    shutil.copytree(os.path.join(REF, "src"), work / "src")
    shutil.copy(os.path.join(REF, "hamilton.cabal"), work / "hamilton.cabal")
    patch = os.path.join(ROOT, "bindings", "haskell", "hamilton-hip.patch")
    r = subprocess.run(["patch", "-p1", "--fuzz=0", "-i", patch], cwd=work, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    src = open(work / "src" / "Numeric" / "Hamilton.hs").read()
    assert "_sysHip" in src and "rk4StepsBatch" in src and "HIP.traceSystem" in src
    cabal = open(work / "hamilton.cabal").read()
    assert "Numeric.Hamilton.HIP" in cabal and "extra-libraries:  hamk" in cabal
    # purely additive: every line of the reference module survives (two are re-indented in the cabal file only)
    ref_lines = open(os.path.join(REF, "src", "Numeric", "Hamilton.hs")).read().splitlines()
    new_lines = set(src.splitlines())
    assert all(l in new_lines for l in ref_lines)


def test_shim_declares_every_entry_point_it_needs():
    """Each `foreign import` of the shim names a function include/hamk.h declares."""
    import re
    hs = open(os.path.join(ROOT, "bindings", "haskell", "Numeric", "Hamilton", "HIP.hs")).read()
    header = open(os.path.join(ROOT, "include", "hamk.h")).read()
    names = re.findall(r'foreign import ccall (?:safe |unsafe )?"&?(hamk_\w+)"', hs)
    assert len(names) >= 20
    for n in names:
        assert re.search(r"\b" + n + r"\s*\(", header), n


# ---------------------------------------------------------------------------------------------------------------------
# No GHC in this image: the shim cannot be compiled.  What CAN be checked without it is that every `foreign import`
# agrees with the C prototype it binds -- arity, and per argument the class of C type (int32 / int64 / uint64 / double /
# pointer / C string) -- and that the hand-written Storable instances use the offsets the C compiler gives the structs.
# ---------------------------------------------------------------------------------------------------------------------
def _c_prototypes():
    import re
    header = open(os.path.join(ROOT, "include", "hamk.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(const\s+char\s*\*|int|void|int64_t)\s+(hamk_\w+)\s*\(([^;{]*?)\)\s*;", header, re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (" ".join(ret.split()), params)
    return protos


def _c_class(decl: str) -> str:
    d = decl.replace("const ", "").strip()
    if "*" in d:
        return "cstring" if d.startswith("char") else "ptr"
    base = d.split()[0]
    return {"int32_t": "i32", "int": "i32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "long": "i64"}[base]


def _hs_class(t: str) -> str:
    t = t.strip()
    if t in ("CString",):
        return "cstring"
    if t.startswith("Ptr") or t.startswith("FunPtr"):
        return "ptr"
    return {"Int32": "i32", "CInt": "i32", "Int64": "i64", "Word64": "u64", "Double": "f64"}[t]


def _split_arrows(sig: str):
    out, depth, cur = [], 0, ""
    i = 0
    while i < len(sig):
        c = sig[i]
        if c == "(":
            depth += 1
        elif c == ")":
            depth -= 1
        if depth == 0 and sig[i:i + 2] == "->":
            out.append(cur.strip())
            cur = ""
            i += 2
            continue
        cur += c
        i += 1
    out.append(cur.strip())
    return out


def _hs_imports():
    import re
    hs = open(os.path.join(ROOT, "bindings", "haskell", "Numeric", "Hamilton", "HIP.hs")).read()
    pat = r'foreign import ccall\s+(?:safe\s+|unsafe\s+)?"(&?)(hamk_\w+)"\s*\n?\s*\w+\s*::(.*?)(?=\nforeign import|\n\n|\n--|\ninstance|\ndata )'
    return [(m.group(1) == "&", m.group(2), " ".join(m.group(3).split())) for m in re.finditer(pat, hs, re.S)], hs


def test_every_foreign_import_matches_its_c_prototype():
    protos = _c_prototypes()
    imports, _ = _hs_imports()
    assert len(imports) >= 20
    for by_address, name, sig in imports:
        assert name in protos, f"{name}: not declared in include/hamk.h"
        ret, params = protos[name]
        if by_address:                                       # FunPtr (Ptr a -> IO ()): a finaliser -- one pointer argument; the C
            assert sig.startswith("FunPtr (") and sig.endswith("-> IO ())"), (name, sig)      # function returns void, or an int status
            assert ret in ("void", "int") and len(params) == 1 and _c_class(params[0]) == "ptr", (name, ret, params)     # nobody reads
            continue
        parts = _split_arrows(sig)
        args, res = parts[:-1], parts[-1]
        assert len(args) == len(params), f"{name}: {len(args)} Haskell arguments, {len(params)} in the C prototype"
        for k, (h, c) in enumerate(zip(args, params)):
            assert _hs_class(h) == _c_class(c), f"{name}: argument {k}: Haskell `{h}` against C `{c}`"
        assert res.startswith("IO "), (name, res)
        want = {"int": "i32", "const char *": "cstring", "const char*": "cstring", "int64_t": "i64"}[ret]
        assert _hs_class(res[3:]) == want, f"{name}: result `{res}` against C `{ret}`"


def _c_layout(tmp_path):
    """sizeof / offsetof of hamk_options and hamk_op as THIS C compiler lays them out."""
    import json
    import re
    header = open(os.path.join(ROOT, "include", "hamk.h")).read()
    body = header[header.index("typedef struct hamk_options {"):header.index("} hamk_options;")]
    body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
    fields = [m.group(1) for m in re.finditer(r"\b(?:u?int32_t|int64_t)\s+(\w+)\s*(?:\[\d+\])?\s*;", body)]
    src = tmp_path / "layout.c"
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "hamk.h"', "int main(void) {",
            '  printf("{\\"sizeof_options\\": %zu, \\"sizeof_op\\": %zu, \\"version_value\\": %u", sizeof(hamk_options), sizeof(hamk_op), (unsigned)HAMK_OPTIONS_VERSION);']
    for f in fields:
        prog.append(f'  printf(", \\"{f}\\": %zu", offsetof(hamk_options, {f}));')
    for f in ("op", "a", "b", "c"):
        prog.append(f'  printf(", \\"op.{f}\\": %zu", offsetof(hamk_op, {f}));')
    prog += ['  printf("}\\n");', "  return 0;", "}"]
    src.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    return json.loads(subprocess.check_output([str(exe)], text=True)), fields


def test_storable_instances_use_the_c_compilers_offsets(tmp_path):
    import re
    lay, fields = _c_layout(tmp_path)
    _, hs = _hs_imports()
    assert lay["sizeof_options"] == 128 and lay["sizeof_op"] == 24
    inst = hs[hs.index("instance Storable Options where"):]
    inst = inst[:inst.index("foreign import")]
    assert int(re.search(r"sizeOf _ = (\d+)", inst).group(1)) == lay["sizeof_options"]
    assert int(re.search(r"optionsVersion = (0x[0-9a-fA-F]+)", hs).group(1), 16) == lay["version_value"]
    # the record's fields, in declaration order, and the C fields they stand for
    rec = hs[hs.index("data Options = Options"):hs.index("defaultOptions ::")]
    rec = re.sub(r"--.*", "", rec)
    hs_fields = re.findall(r"\bopt[A-Z]\w*", rec)
    def snake(name):                                         # optRk4MinWaves -> rk4_min_waves
        return re.sub(r"(?<!^)(?=[A-Z])", "_", name[3:]).lower()
    c_names = [snake(f) for f in hs_fields]
    settable = [f for f in fields if f not in ("size", "version", "_align", "reserved")]
    assert c_names == settable, (c_names, settable)
    # peek: one offset per field, in order
    peek = inst[inst.index("peek p ="):inst.index("where f = peekByteOff p")]
    offs = [int(x) for x in re.findall(r"\bf (\d+)", peek)] + [int(x) for x in re.findall(r"peekByteOff p (\d+)", peek)]
    assert offs == [lay[f] for f in settable], (offs, [lay[f] for f in settable])
    # poke: size at 0, version at its offset, the Int32 fields as an arithmetic progression from `mapping`, ensemble_size last
    poke = inst[inst.index("poke p"):]
    assert re.search(r"pokeByteOff p 0 \(128 :: Word32\)", poke)
    assert re.search(rf"pokeByteOff p {lay['version']} optionsVersion", poke)
    m = re.search(r"zip \[(\d+), (\d+) \.\.\] \[([^\]]*)\]", poke)
    first, second, names = int(m.group(1)), int(m.group(2)), [x.strip() for x in m.group(3).split(",")]
    int32_fields = [f for f in settable if f != "ensemble_size"]
    assert len(names) == len(int32_fields)
    assert [first + k * (second - first) for k in range(len(names))] == [lay[f] for f in int32_fields]
    assert re.search(rf"pokeByteOff p {lay['ensemble_size']} ens", poke)
    # hamk_op: { int32 op, a, b, pad; double c }
    op = hs[hs.index("instance Storable Op where"):hs.index("foreign import")]
    assert int(re.search(r"sizeOf _ = (\d+)", op).group(1)) == lay["sizeof_op"]
    assert [int(x) for x in re.findall(r"peekByteOff p (\d+)", op)] == [lay["op.op"], lay["op.a"], lay["op.b"], lay["op.c"]]
