"""CPU: pin the C oracle against the independently derived high-precision fixtures.

The reference has no golden vectors for this path (test/Spec.hs:1-2); the fixtures
come from oracle/gen_golden.py (sympy + 50-digit mpmath).  Tolerance ladder T1 of
SURVEY.md section 8c: 1e-12 * max(1, |y|), loosened by cond(K) where K is ill-conditioned.
"""
import numpy as np
import pytest

from conftest import ALL_GOLDEN_SYSTEMS, BYHAND_SYSTEMS, CHAIN_GOLDEN_SYSTEMS, REFERENCE_SYSTEMS, fvec, load_golden
from hamilton_amd import examples as E

T1 = 1e-12


def tol_for(pt, *vals):
    scale = max([1.0] + [float(np.max(np.abs(v))) for v in vals])
    cond = max(1.0, float(pt["cond_hint"]))
    return T1 * scale * max(1.0, cond / 1e3)


@pytest.fixture(scope="module")
def systems(oracle_lib):
    return {name: oracle_lib.OracleSystem(E.get(name)) for name in ALL_GOLDEN_SYSTEMS}


@pytest.mark.parametrize("name", ALL_GOLDEN_SYSTEMS)
def test_state_functions_match_golden(systems, name):
    o, g = systems[name], load_golden(name)
    assert (g["m"], g["n"]) == (o.m, o.n)
    for pt in g["points"]:
        q, qd, p = fvec(pt["q"]), fvec(pt["qd"]), fvec(pt["p"])
        tol = tol_for(pt, p, fvec(pt["dp"]))
        np.testing.assert_allclose(o.coords(q), fvec(pt["x"]), rtol=0, atol=tol)
        np.testing.assert_allclose(o.jacobian(q), np.array([fvec(r) for r in pt["jac"]]), rtol=0, atol=tol)
        np.testing.assert_allclose(o.momenta(q, qd), p, rtol=0, atol=tol)                  # Hamilton.hs:262-269
        np.testing.assert_allclose(o.velocities(q, p), fvec(pt["vel"]), rtol=0, atol=tol)  # :316-324
        assert abs(o.keC(q, qd) - float(pt["keC"])) <= tol                                  # :288-296
        assert abs(o.keP(q, p) - float(pt["keP"])) <= tol                                   # :341-349
        assert abs(o.pe(q) - float(pt["pe"])) <= tol                                        # :182-186
        assert abs(o.lagrangian(q, qd) - float(pt["lagrangian"])) <= tol                    # :301-309
        assert abs(o.hamiltonian(q, p) - float(pt["hamiltonian"])) <= tol                   # :353-361


@pytest.mark.parametrize("name", ALL_GOLDEN_SYSTEMS)
def test_hameqs_matches_golden(systems, name):
    o, g = systems[name], load_golden(name)
    for pt in g["points"]:
        q, p = fvec(pt["q"]), fvec(pt["p"])
        dq, dp = o.hameqs(q, p)                                                             # :370-387
        tol = tol_for(pt, fvec(pt["dq"]), fvec(pt["dp"]))
        np.testing.assert_allclose(dq, fvec(pt["dq"]), rtol=0, atol=tol)
        np.testing.assert_allclose(dp, fvec(pt["dp"]), rtol=0, atol=tol)


@pytest.mark.parametrize("name", CHAIN_GOLDEN_SYSTEMS)
def test_oracle_matches_the_closed_form_chain_fixtures(oracle_lib, name):
    """BASELINE config 5 (N = 8, 16, 32): the oracle's tape interpreter + literal Hamilton.hs:262-387 against 50-digit values
    of the chain's hand-derived mechanics (closed-form mass matrix, Hamilton's equations written out; no tape, no AD)."""
    o, g = oracle_lib.OracleSystem(E.get(name)), load_golden(name)
    assert (g["m"], g["n"]) == (o.m, o.n) and len(g["points"]) == 13
    for pt in g["points"]:
        q, qd, p = fvec(pt["q"]), fvec(pt["qd"]), fvec(pt["p"])
        tol = tol_for(pt, p, fvec(pt["dp"]), fvec(pt["dq"]))
        np.testing.assert_allclose(o.coords(q), fvec(pt["x"]), rtol=0, atol=tol)
        np.testing.assert_allclose(o.momenta(q, qd), p, rtol=0, atol=tol)
        np.testing.assert_allclose(o.velocities(q, p), fvec(pt["vel"]), rtol=0, atol=tol)
        for f, key, args in ((o.keC, "keC", (q, qd)), (o.keP, "keP", (q, p)), (o.pe, "pe", (q,)),
                             (o.lagrangian, "lagrangian", (q, qd)), (o.hamiltonian, "hamiltonian", (q, p))):
            assert abs(f(*args) - float(pt[key])) <= tol, (name, key)
        dq, dp = o.hameqs(q, p)
        np.testing.assert_allclose(dq, fvec(pt["dq"]), rtol=0, atol=tol)
        np.testing.assert_allclose(dp, fvec(pt["dp"]), rtol=0, atol=tol)


@pytest.mark.parametrize("name", BYHAND_SYSTEMS)
def test_oracle_matches_the_by_hand_fixtures(oracle_lib, name):
    """The reference's own systems (app/Examples.hs:61-162, README.md:88-103) against 50-digit values of mass matrix, potential and
    Hamilton's equations WRITTEN OUT BY HAND (oracle/gen_golden_byhand.py: nothing imported from hamilton_amd, no tape, no AD) -- the
    only fixtures a transcription error in hamilton_amd/examples.py cannot hide behind."""
    o, g = oracle_lib.OracleSystem(E.get(name)), load_golden("byhand:" + name)
    assert len(g["points"]) >= 13
    for pt in g["points"]:
        q, qd, p = fvec(pt["q"]), fvec(pt["qd"]), fvec(pt["p"])
        assert len(q) == o.n and len(pt["x"]) == o.m
        tol = tol_for(pt, p, fvec(pt["dp"]), fvec(pt["dq"]))
        np.testing.assert_allclose(o.coords(q), fvec(pt["x"]), rtol=0, atol=tol)
        np.testing.assert_allclose(o.momenta(q, qd), p, rtol=0, atol=tol)
        np.testing.assert_allclose(o.velocities(q, p), fvec(pt["vel"]), rtol=0, atol=tol)
        for f, key, args in ((o.keC, "keC", (q, qd)), (o.keP, "keP", (q, p)), (o.pe, "pe", (q,)),
                             (o.lagrangian, "lagrangian", (q, qd)), (o.hamiltonian, "hamiltonian", (q, p))):
            assert abs(f(*args) - float(pt[key])) <= tol, (name, key)
        dq, dp = o.hameqs(q, p)
        np.testing.assert_allclose(dq, fvec(pt["dq"]), rtol=0, atol=tol)
        np.testing.assert_allclose(dp, fvec(pt["dp"]), rtol=0, atol=tol)


@pytest.mark.parametrize("name", [n for n in BYHAND_SYSTEMS if n != "doublePendulumReadme"])
def test_by_hand_and_symbolic_fixtures_agree(name):
    """Two derivations that share no code -- sympy over the Python restatement (gen_golden.py) and the hand-written mechanics
    (gen_golden_byhand.py) -- at the same points, digit for digit (both files carry 30 digits; 1e-25 leaves room for the last ones)."""
    import mpmath as mp
    mp.mp.dps = 40
    a, b = load_golden(name)["points"], load_golden("byhand:" + name)["points"]
    assert len(a) == len(b)
    for pa, pb in zip(a, b):
        assert pa["q"] == pb["q"] and pa["qd"] == pb["qd"]
        for key in ("p", "x", "vel", "dq", "dp", "keC", "keP", "pe", "lagrangian", "hamiltonian"):
            va = pa[key] if isinstance(pa[key], list) else [pa[key]]
            vb = pb[key] if isinstance(pb[key], list) else [pb[key]]
            for u, w in zip(va, vb):
                assert abs(mp.mpf(u) - mp.mpf(w)) <= mp.mpf(10) ** -25 * (1 + abs(mp.mpf(u))), (name, key, u, w)


def test_two_body_angle_is_cyclic_in_the_by_hand_fixture():
    """twoBody's potential depends on r only (app/Examples.hs:138): dp_theta = 0 exactly, in the fixture and in the oracle."""
    for pt in load_golden("byhand:twoBody")["points"]:
        assert float(pt["dp"][1]) == 0.0


def test_hessian_layout_is_dJ_dqi(systems):
    """`_sysHessian q !! i` = dJ/dq_i (Hamilton.hs:222, :227-233): check by central differences of J."""
    o = systems["spring"]
    q = np.array([0.3, -0.1, 0.4])
    H = o.hessian(q)
    eps = 1e-6
    for i in range(o.n):
        e = np.zeros(o.n); e[i] = eps
        fd = (o.jacobian(q + e) - o.jacobian(q - e)) / (2 * eps)
        np.testing.assert_allclose(H[i], fd, rtol=0, atol=1e-8)


def test_double_pendulum_initial_rhs(systems):
    """seInit of doublePendulum 1 1 is q=(pi/2,0), p=(0,0); hamEqs there is ((0,0),(-10,0))."""
    o = systems["doublePendulum"]
    spec = E.get("doublePendulum")
    p = o.momenta(spec.q0, spec.qd0)
    np.testing.assert_array_equal(p, [0.0, 0.0])
    dq, dp = o.hameqs(spec.q0, p)
    np.testing.assert_allclose(dq, [0, 0], atol=1e-15)
    np.testing.assert_allclose(dp, [-10, 0], atol=1e-14)


def test_readme_worked_example(systems, oracle_lib):
    """README.md:92-126: masses (1,1,2,2), g=5 double pendulum, config0 = Cfg (1,0) (0,0.5)."""
    o = oracle_lib.OracleSystem(E.double_pendulum(1.0, 2.0))
    q, qd = np.array([1.0, 0.0]), np.array([0.0, 0.5])
    p = o.momenta(q, qd)
    # K = [[m1+m2, m2/2 cos(t1-t2)], [m2/2 cos(t1-t2), m2/4]] for unit/half link lengths
    K = np.array([[3.0, np.cos(1.0)], [np.cos(1.0), 0.5]])
    np.testing.assert_allclose(p, K @ qd, rtol=1e-14)
    np.testing.assert_allclose(o.velocities(q, p), qd, rtol=1e-13, atol=1e-15)
    assert abs(o.keC(q, qd) - 0.5 * qd @ K @ qd) < 1e-15


@pytest.mark.parametrize("name", REFERENCE_SYSTEMS)
def test_invariants(systems, name):
    """fromPhase . toPhase = id; keC = keP . toPhase (Hamilton.hs:279-296, :332-349)."""
    o, spec = systems[name], E.get(name)
    q, qd = E.sample_config(spec, 100, 16)
    p = o.to_phase_batch(q, qd)
    back, st = o.from_phase_batch(q, p)
    assert not st.any()
    np.testing.assert_allclose(back, qd, rtol=1e-11, atol=1e-12)
    keC, _ = o.observe_config_batch(q, qd)
    keP, _, _ = o.observe_batch(q, p)
    np.testing.assert_allclose(keC, keP, rtol=1e-11, atol=1e-13)


def test_two_body_theta_is_cyclic(systems):
    """twoBody's potential depends on r only (Examples.hs:138) => dp_theta/dt == 0."""
    o, spec = systems["twoBody"], E.get("twoBody")
    q, qd = E.sample_config(spec, 0, 32)
    p = o.to_phase_batch(q, qd)
    _, dp, _ = o.hameqs_batch(q, p)
    assert np.max(np.abs(dp[1])) < 1e-13


@pytest.mark.parametrize("name", REFERENCE_SYSTEMS + ["threeBodyPolar"])
def test_trajectories_against_taylor_truth(systems, name):
    """Integrators vs mpmath.odefun truth: RKF45 within its tolerance, RK4 within O(dt^4)."""
    o, g = systems[name], load_golden(name)
    tr = g["trajectory"]
    q0, p0 = fvec(tr["q0"]), fvec(tr["p0"])
    spec = E.get(name)
    for st in tr["states"]:
        t = float(st["t"])
        qt, pt = fvec(st["q"]), fvec(st["p"])
        scale = max(1.0, np.max(np.abs(qt)), np.max(np.abs(pt)))
        qo, po = o.evolve_ham(q0, p0, [0.0, t])                 # evolveHam semantics, eps 1.49e-8
        err = max(np.max(np.abs(qo[1] - qt)), np.max(np.abs(po[1] - pt)))
        assert err < 2e-6 * scale, (name, t, err)
        nst = int(round(t / spec.dt))
        qr, pr = o.rk4_steps(q0, p0, spec.dt, nst)
        err4 = max(np.max(np.abs(qr - qt)), np.max(np.abs(pr - pt)))
        assert err4 < 5e-5 * scale, (name, t, err4)


def test_gsl_step_accounting(systems):
    """stepHam 0.01 on the double pendulum from seInit: 4 accepted sub-steps
    (1e-4 -> 5e-4 -> 2.5e-3 -> remainder), no rejects.  Old gsl_odeiv evaluates dydt_in at the top of
    every evolve_apply: 4 x 7 = 28 RHS evaluations; gsl_odeiv2 re-uses
    dydt_out of the previous step: 1 + 4 x 6 = 25.  Same states either way (one interval)."""
    o, spec = systems["doublePendulum"], E.get("doublePendulum")
    res = {}
    for api, want in ((1, [28, 4, 0, 0]), (2, [25, 4, 0, 0])):
        o.gsl_api = api
        counts = []
        res[api] = o.step_ham(0.01, spec.q0, [0.0, 0.0], counts)
        assert counts == want, (api, counts)
    o.gsl_api = 2
    np.testing.assert_array_equal(res[1][0], res[2][0]); np.testing.assert_array_equal(res[1][1], res[2][1])


def test_evolve_ham_rows_and_carry(systems):
    """Row 0 is the initial state (Hamilton.hs:443-462); h carries across output times, so a
    multi-time call differs from restarted stepHam calls only at truncation level."""
    o, spec = systems["doublePendulum"], E.get("doublePendulum")
    q0, p0 = np.array(spec.q0), np.array([0.0, 0.0])
    ts = np.array([0.0, 0.05, 0.1, 0.2])
    qo, po = o.evolve_ham(q0, p0, ts)
    np.testing.assert_array_equal(qo[0], q0)
    np.testing.assert_array_equal(po[0], p0)
    q, p = q0, p0
    for r in range(1, len(ts)):
        q, p = o.step_ham(ts[r] - ts[r - 1], q, p)
        assert np.max(np.abs(q - qo[r])) < 1e-6


def test_rk4_vs_stepham_one_step_is_truncation_level(systems):
    """Tolerance ladder T4: one RK4 step vs stepHam (RKF45) <= 1e-8 for dt = 0.01."""
    o, spec = systems["doublePendulum"], E.get("doublePendulum")
    q0, p0 = np.array(spec.q0), np.array([0.0, 0.0])
    qa, pa = o.rk4_steps(q0, p0, 0.01, 1)
    qb, pb = o.step_ham(0.01, q0, p0)
    assert max(np.max(np.abs(qa - qb)), np.max(np.abs(pa - pb))) < 1e-8


def test_energy_conservation_rkf45(systems):
    o, spec = systems["twoBody"], E.get("twoBody")
    q0 = np.array(spec.q0); p0 = o.momenta(q0, spec.qd0)
    h0 = o.hamiltonian(q0, p0)
    qo, po = o.evolve_ham(q0, p0, np.linspace(0, 5, 6))
    for r in range(6):
        assert abs(o.hamiltonian(qo[r], po[r]) - h0) < 1e-6 * abs(h0)
