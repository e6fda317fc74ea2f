"""Round 6: the generator's SYMBOLIC mass matrix (hamk_codegen.cpp symbolic_mass_matrix).  Where the coordinate map is a polynomial in its
inputs and in sincos of polynomial arguments, K = J^T M J (Hamilton.hs:380) is derived as polynomials, sin^2 + cos^2 = 1 is applied, and
dT/dq = -1/2 v^T (dK/dq) v (Hamilton.hs:382-385) follows from it -- no second-order sweep.  CPU checks: which systems get it, what it looks
like for the reference's examples, and its VALUES: the emitted expressions are evaluated here (numpy) against the oracle's J^T M J and
against central differences of themselves.  The kernels that use it are compared with the oracle in the host emulation and on the GPU."""
import re

import numpy as np
import pytest

from hamilton_amd import examples as E


def sym_functions(src):
    """{"K": {(a, b): python expression}, "dT": {i: expression}} parsed from a generated module (None where the module has none)."""
    def body(name):
        m = re.search(r"static void %s\(const double \(&q\)\[N\].*?\{\n(.*?)\n  \}" % name, src, re.S)
        return m.group(1) if m else None
    hexf = lambda e: re.sub(r"\(?(-?0x[0-9a-f.]+p[+-]\d+)\)?", lambda m: repr(float.fromhex(m.group(1))), e)
    fix = lambda e: hexf(e).replace("tc.s[", "s[").replace("tc.c[", "c[")
    out = {"K": None, "dT": None}
    kb, db = body("mass_matrix_sym"), body("dT_sym")
    if kb:
        out["K"] = {}
        for a, b, e in re.findall(r"K\[(\d+)\]\[(\d+)\] = ([^;]*);", kb):
            if not e.startswith("K["):
                out["K"][(int(a), int(b))] = fix(e)
    if db:
        out["dT"] = {int(i): fix(e) for i, e in re.findall(r"dT\[(\d+)\] = ([^;]*);", db)}
    return out


def trig_slots(src, spec):
    """operand of every trig-cache slot as a function of q: the slots of these systems are sincos of INPUTS (trig_input table)."""
    m = re.search(r"trig_input\(int slot\) \{\n\s*constexpr int w\[\d+\] = \{([^}]*)\}", src)
    return [int(t) for t in m.group(1).split(",")]


@pytest.fixture(scope="module")
def api(hamk_lib):
    from hamilton_amd import api as _api
    return _api


@pytest.mark.parametrize("name,sym_k,sym_dt", [("pendulum", True, True), ("doublePendulum", True, True), ("room", True, True), ("twoBody", True, True),
                                               ("spring", True, True), ("threeBodyPolar", True, True), ("chain4", True, False), ("chain6", True, False),
                                               ("bezier", False, False), ("opcodeZoo", False, False), ("chain8", False, False)])
def test_which_systems_get_a_symbolic_mass_matrix(api, name, sym_k, sym_dt):
    """The reference's trigonometric examples and BASELINE configs 2-4: K and dT/dq symbolic; small chains: K only (their dT/dq has n (n - 1)
    quartic terms: the directional sweep stays); bezier (the symbolic K is longer than the numerical sum), opcodeZoo (not a polynomial) and
    everything beyond n = 7: none."""
    src = api.system_from_spec(E.get(name)).source
    assert ("HAS_SYM_K = true" in src) == sym_k and ("HAS_SYM_DT = true" in src) == sym_dt


def test_the_examples_simplify_to_their_textbook_form(api):
    f = sym_functions(api.system_from_spec(E.get("doublePendulum")).source)
    assert eval(f["K"][(0, 0)]) == 2.0 and eval(f["K"][(1, 1)]) == 0.25                       # m1 + m2, m2 / 4 (Examples.hs:75-94, unit masses)
    f = sym_functions(api.system_from_spec(E.get("twoBody")).source)
    mu = 5.0 * (0.5 / 5.5) ** 2 + 0.5 * (5.0 / 5.5) ** 2
    assert abs(eval(f["K"][(0, 0)]) - mu) < 1e-16 and f["K"][(0, 1)] == "0.0" and f["dT"][1] == "0.0"      # theta is cyclic (Examples.hs:138)
    f = sym_functions(api.system_from_spec(E.get("threeBodyPolar")).source)
    assert all(f["K"][(a, b)] == "0.0" for a in range(6) for b in range(a + 1, 6))            # polar coordinates: K = diag(1, r^2, ...)
    assert [f["K"][(i, i)] for i in range(6)] == ["1.0", "q[0] * q[0]", "1.0", "q[2] * q[2]", "1.0", "q[4] * q[4]"]


@pytest.mark.parametrize("name", ["pendulum", "doublePendulum", "doublePendulumReadme", "twoBody", "spring", "threeBodyPolar", "room", "chain4", "doublePendulum~mixed"])
def test_symbolic_values_against_the_oracle(api, oracle_lib, name):
    spec = E.get(name)
    src = api.system_from_spec(spec).source
    f = sym_functions(src)
    assert f["K"] is not None
    slots = trig_slots(src, spec)
    o = oracle_lib.OracleSystem(spec)
    qs, qds = E.sample_config(spec, 17, 12)
    n = spec.n

    def K_at(q):
        s, c = [np.sin(q[j]) for j in slots], [np.cos(q[j]) for j in slots]
        K = np.zeros((n, n))
        for (a, b), e in f["K"].items():
            K[a, b] = K[b, a] = eval(e, {"q": q, "s": s, "c": c})
        return K

    for i in range(qs.shape[1]):
        q, v = qs[:, i].copy(), qds[:, i] + 0.37
        J = o.jacobian(q)
        Kref = J.T @ np.diag(spec.inertia) @ J                                                 # Hamilton.hs:380
        np.testing.assert_allclose(K_at(q), Kref, rtol=0, atol=4e-15 * max(1.0, np.abs(Kref).max()))
        if f["dT"] is not None:
            s, c = [np.sin(q[j]) for j in slots], [np.cos(q[j]) for j in slots]
            for k, e in f["dT"].items():
                h = 1e-6
                qp, qm = q.copy(), q.copy()
                qp[k] += h; qm[k] -= h
                fd = -0.5 * v @ ((K_at(qp) - K_at(qm)) / (2 * h)) @ v                          # -1/2 v^T (dK/dq_k) v
                got = eval(e, {"q": q, "s": s, "c": c, "v": v})
                assert abs(got - fd) <= 1e-8 * max(1.0, abs(fd)), (name, k, got, fd)
            # and through the reference's own formula: dp = -(dT/dq + dU/dq) with v = K^-1 p (Hamilton.hs:382-387)
            p = Kref @ v
            _, dp = o.hameqs(q, p)
            gU = np.array([(o.pe(q + h * np.eye(n)[k]) - o.pe(q - h * np.eye(n)[k])) / (2 * h) for k in range(n)])
            dT = np.array([eval(f["dT"][k], {"q": q, "s": s, "c": c, "v": v}) for k in range(n)])
            np.testing.assert_allclose(-(dT + gU), dp, rtol=0, atol=2e-7 * max(1.0, np.abs(dp).max()))
