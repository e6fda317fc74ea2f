"""The host shims' recorders produce BYTE-IDENTICAL tapes: what crosses the C ABI for a system is a
function of the system, not of the host language.  The C++ mirror (include/hamilton.hpp) defines
the six example systems of the reference independently (tests/cpp/reference_systems.cpp, following
app/Examples.hs) and dumps its tapes; the Python recorder's (hamilton_amd/tracer.py over
hamilton_amd/examples.py) must match them to the byte, and the hand-written C tape of
tests/c/abi_smoke.c (a pendulum, no recorder at all) must be what both recorders emit for it.
Both recorders ship the canonical form (depth-first post-order from the outputs), so emission order
of constants, dead operands of folded expressions and the like cannot leak into the tape."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import REFERENCE_SYSTEMS, ROOT
from hamilton_amd import examples as E
from hamilton_amd.tracer import HamkOp


@pytest.fixture(scope="module")
def cpp_tapes(hamk_lib, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "reference_systems")
    libdir = os.path.join(ROOT, "hamilton_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "reference_systems.cpp"), "-o", out,
                           "-L" + libdir, "-lhamk", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    tapes = {}
    for line in subprocess.check_output([out], text=True).splitlines():
        head, body = line.split(" :")
        name, which, *outs = head.split()
        tapes[(name, which)] = ([int(o) for o in outs], bytes.fromhex("".join(body.split())))
    return tapes


def tape_bytes(t):
    arr, n, _ = t.as_ctypes()
    return bytes(ctypes.string_at(arr, n * ctypes.sizeof(HamkOp)))


@pytest.mark.parametrize("name", REFERENCE_SYSTEMS)
def test_python_and_cpp_recorders_emit_identical_tapes(cpp_tapes, name):
    tf, tu = E.get(name).trace()
    for which, t in (("f", tf), ("u", tu)):
        outs, raw = cpp_tapes[(name, which)]
        assert outs == list(t.outs), (name, which, outs, t.outs)
        assert len(raw) == 24 * len(t), (name, which, len(raw) // 24, len(t))
        assert raw == tape_bytes(t), (name, which)


def test_hand_written_c_tape_is_what_the_recorders_emit():
    """tests/c/abi_smoke.c spells its pendulum tape out by hand; parse the initialisers and compare
    with the recorders' canonical tape of the same function (sin q, 0.5 - cos q; U = x1)."""
    src = open(os.path.join(ROOT, "tests", "c", "abi_smoke.c")).read()
    ops = re.findall(r"= op\((HAMK_OP_\w+),\s*(-?\d+),\s*(-?\d+),\s*([-0-9.e]+)\);", src)
    assert ops, "no hamk_op initialisers found in abi_smoke.c"
    from hamilton_amd import tracer as T
    code = {"HAMK_OP_CONST": T.OP_CONST, "HAMK_OP_INPUT": T.OP_INPUT, "HAMK_OP_SIN": T.OP_SIN, "HAMK_OP_COS": T.OP_COS,
            "HAMK_OP_SUB": T.OP_SUB, "HAMK_OP_ADD": T.OP_ADD, "HAMK_OP_MUL": T.OP_MUL, "HAMK_OP_NEG": T.OP_NEG}
    hand = [(code[o], int(a), int(b), float(c)) for o, a, b, c in ops]
    tf, tu = E.get("pendulum").trace()
    assert hand[:len(tf)] == [tuple(x) for x in tf.ops], (hand[:len(tf)], tf.ops)
    assert hand[len(tf):len(tf) + len(tu)] == [tuple(x) for x in tu.ops]
