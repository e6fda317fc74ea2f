"""CPU: the library's DEVICE code on the host.  tests/host_emulation/hip_shim.hpp supplies just
enough of the HIP device environment to compile hamilton_amd/csrc/hamk_device.hpp plus a generated
system with g++; the kernels are then run thread by thread and compared with the oracle.  Covers,
without a GPU: the tape -> C++ generator, the jets of every opcode, the three AD strategies (incl.
the generated reverse sweep), the solves, sincos_f64 / incremental sincos, the RK4 body and the
GSL-semantics RKF45 body with its controller.  TEST INFRASTRUCTURE: nothing here is linked into
libhamk.so and the product has no CPU path; what this cannot see is the GPU compiler."""
import ctypes
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import ALL_GOLDEN_SYSTEMS, ROOT
from hamilton_amd import examples as E

EMU = os.path.join(ROOT, "tests", "host_emulation")
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
P = lambda a: a.ctypes.data_as(_dp)
I = lambda a: a.ctypes.data_as(_ip)
LL = ctypes.c_longlong


@pytest.fixture(scope="module")
def emulate(hamk_lib, tmp_path_factory):
    from hamilton_amd import api
    cache = {}
    tmp = tmp_path_factory.mktemp("emu")

    def make(spec, env=None):
        old = {}
        for k, v in (env or {}).items():
            old[k] = os.environ.get(k)
            os.environ[k] = v
        try:
            s = api.system_from_spec(spec)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        src = s.source
        assert "hamk_device.hpp" in src, "lane path expected (wave kernels are not emulated)"
        key = hashlib.sha1(src.encode()).hexdigest()[:16]
        if key not in cache:
            cpp, so = str(tmp / f"{key}.cpp"), str(tmp / f"{key}.so")
            with open(cpp, "w") as fh:
                fh.write(src + open(os.path.join(EMU, "driver.inc")).read())
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-fno-gnu-unique", "-Wno-unknown-pragmas", "-Wno-attributes",
                                   "-include", os.path.join(EMU, "hip_shim.hpp"), "-I" + os.path.join(ROOT, "hamilton_amd", "csrc"),
                                   "-o", so, cpp])
            cache[key] = ctypes.CDLL(so)
        return cache[key], src
    return make


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(1.0, np.abs(np.asarray(b)))))


def check_against_oracle(L, spec, o, B=48, start=11, tol=1e-11, steps=3, dt_ham=None, qd_kick=0.0):
    q, qd = E.sample_config(spec, start, B)
    if qd_kick:                                             # (the chains' sampling box has qd = 0: p = 0 would make every solve trivial)
        qd = qd + qd_kick * np.cos(1.0 + np.arange(spec.n * B, dtype=np.float64).reshape(spec.n, B))
    p = o.to_phase_batch(q, qd)
    st = np.zeros(B, np.int32)
    got = np.zeros_like(q)
    L.emu_to_phase(P(q), P(qd), P(got), LL(B))
    assert relerr(got, p) < tol
    odq, odp, ost = o.hameqs_batch(q, p)
    good = ost == 0
    cond = np.array([np.linalg.cond(o.jacobian(q[:, i]).T @ np.diag(spec.inertia) @ o.jacobian(q[:, i])) for i in range(B)])
    scale = np.maximum(1.0, cond / 100.0)
    dq, dp = np.zeros_like(q), np.zeros_like(q)
    L.emu_hameqs(P(q), P(p), P(dq), P(dp), LL(B), I(st))
    err = np.maximum(np.abs(dq - odq).max(0) / np.maximum(1.0, np.abs(odq).max(0)),
                     np.abs(dp - odp).max(0) / np.maximum(1.0, np.abs(odp).max(0)))
    assert np.all(err[good] <= tol * scale[good]), float(np.max(err[good] / scale[good]))
    ke, pe, h = np.zeros(B), np.zeros(B), np.zeros(B)
    L.emu_observe(P(q), P(p), P(ke), P(pe), P(h), LL(B), I(st))
    oke, ope, oh = o.observe_batch(q, p)
    assert relerr(pe, ope) < tol and np.all((np.abs(h - oh) / np.maximum(1.0, np.abs(oh)))[good] <= tol * scale[good])
    x = np.zeros((spec.m, B))
    L.emu_coords(P(q), P(x), LL(B))
    assert relerr(x, o.coords_batch(q)) < tol
    v = np.zeros_like(q)
    L.emu_from_phase(P(q), P(p), P(v), LL(B), I(st))
    ov, _ = o.from_phase_batch(q, p)
    assert np.all((np.abs(v - ov).max(0) / np.maximum(1.0, np.abs(ov).max(0)))[good] <= tol * scale[good])
    kec, lag = np.zeros(B), np.zeros(B)
    L.emu_observe_config(P(q), P(qd), P(kec), P(lag), LL(B))
    okec, olag = o.observe_config_batch(q, qd)
    assert relerr(kec, okec) < tol and relerr(lag, olag) < tol
    q2, p2 = q.copy(), p.copy()
    L.emu_rk4(P(q2), P(p2), LL(B), ctypes.c_double(spec.dt), steps, I(st))
    oq, op = o.rk4_steps_batch(q, p, spec.dt, steps)
    e2 = np.maximum(np.abs(q2 - oq).max(0), np.abs(p2 - op).max(0)) / np.maximum(1.0, np.abs(op).max(0))
    assert np.all(e2[good] <= 10 * tol * scale[good]), float(np.max(e2[good] / scale[good]))
    q3, p3, ns = q.copy(), p.copy(), np.zeros(B, np.int32)
    dth = dt_ham if dt_ham is not None else 4 * spec.dt
    L.emu_step_ham(P(q3), P(p3), LL(B), ctypes.c_double(dth), I(st), I(ns))
    sq, sp, sns = o.step_ham_batch(q, p, dth)
    same = (ns == sns) & good
    assert same[good].mean() > 0.9, float(same[good].mean())
    e3 = np.maximum(np.abs(q3 - sq).max(0), np.abs(p3 - sp).max(0)) / np.maximum(1.0, np.abs(sp).max(0))
    assert np.all(e3[same] <= 100 * tol * scale[same]), float(np.max(e3[same] / scale[same]))


@pytest.mark.parametrize("api", [1, 2])
def test_evolveham_time_grid_on_host(emulate, oracle_lib, api):
    """hmatrix-gsl's output loop under both of gsl-ode.c's bindings (hamk.h: hamk_system_set_gsl_api).
    api 1, old gsl_odeiv: `for each ti: while (t < ti) step`, h written back after every accepted
    step; repeated and decreasing times do no stepping.  api 2, gsl_odeiv2 driver (the default):
    h is NOT written back on a final (clipped) step, a repeated time does no stepping.  Row 0 = the
    initial state.  The RKF45 body with a time grid against the oracle's restatement of the same
    binding: identical sub-step counts on every lane, states to roundoff."""
    spec = E.get("doublePendulum")
    o = oracle_lib.OracleSystem(spec)
    o.gsl_api = api
    L, _ = emulate(spec)
    L.emu_set_gsl_api(api)
    try:
        B = 40
        q, qd = E.sample_config(spec, 3, B)
        p = o.to_phase_batch(q, qd)
        ts = np.array([0.0, 0.05, 0.05, 0.02, 0.2, 0.21]) if api == 1 else np.array([0.0, 0.05, 0.05, 0.12, 0.2, 0.21])
        qo, po = np.zeros((len(ts), spec.n, B)), np.zeros((len(ts), spec.n, B))
        st, ns = np.zeros(B, np.int32), np.zeros(B, np.int32)
        L.emu_evolve_ham(P(q), P(p), len(ts), P(ts), P(qo), P(po), LL(B), I(st), I(ns))
        oq, op, ons = o.evolve_ham_batch(q, p, ts)
        assert not st.any() and not o.last_fail.any() and np.array_equal(qo[0], q) and np.array_equal(po[0], p)
        assert np.array_equal(qo[2], qo[1])
        if api == 1:
            assert np.array_equal(qo[3], qo[1])
        assert np.array_equal(ns, ons), (ns[:8], ons[:8])
        assert relerr(qo, oq) < 1e-11 and relerr(po, op) < 1e-11
        # the two bindings are different integrators from the second output time on
        o.gsl_api = 3 - api
        tsm = np.array([0.0, 0.05, 0.12, 0.2, 0.21])
        xq, xp, xns = o.evolve_ham_batch(q, p, tsm)
        o.gsl_api = api
        yq, yp, yns = o.evolve_ham_batch(q, p, tsm)
        assert np.array_equal(xq[1], yq[1]) and not np.array_equal(xq[2], yq[2]) and not np.array_equal(xns, yns)
        assert relerr(xq, yq) < 1e-6                       # ... that agree to the controller's tolerance
    finally:
        L.emu_set_gsl_api(2)


@pytest.mark.parametrize("name", ["doublePendulum", "chain8"])
def test_odeiv2_backward_grid_and_failure_on_host(emulate, oracle_lib, name):
    """(doublePendulum: the unrolled body; chain8: the body with parked stage vectors.)
    gsl_odeiv2 semantics the old API does not have: (a) the direction of integration is the sign
    of the initial step, so a monotone decreasing grid integrates BACKWARDS (the old API's
    `while (t < ti)` does nothing); (b) a step that must shrink but cannot -- here: tolerances no
    fp64 step can meet -- is GSL_FAILURE: the lane stops, ST_UNDERFLOW, later rows = last state."""
    spec = E.get(name)
    o = oracle_lib.OracleSystem(spec)
    L, _ = emulate(spec)
    B = 12
    q, qd = E.sample_config(spec, 5, B)
    if name.startswith("chain"):
        qd = qd + 0.4 * np.cos(np.arange(spec.n * B).reshape(spec.n, B))
    p = o.to_phase_batch(q, qd)
    ts = np.array([0.0, -0.05, -0.12])
    qo, po = np.zeros((3, spec.n, B)), np.zeros((3, spec.n, B))
    st, ns = np.zeros(B, np.int32), np.zeros(B, np.int32)
    L.emu_evolve_ham(P(q), P(p), 3, P(ts), P(qo), P(po), LL(B), I(st), I(ns))
    oq, op, ons = o.evolve_ham_batch(q, p, ts)
    assert not st.any() and np.array_equal(ns, ons) and ns.min() > 2
    assert relerr(qo, oq) < 1e-11 and relerr(po, op) < 1e-11
    fq, fp, _ = o.evolve_ham_batch(oq[2], op[2], np.array([-0.12, 0.0]))          # and forward again: back at the start
    assert relerr(fq[1], q) < 1e-6 and relerr(fp[1], p) < 1e-6
    o.gsl_api = 1
    nq, _, nns = o.evolve_ham_batch(q, p, ts)
    assert np.array_equal(nq[2], q) and not nns.any()                                # old API: no stepping at all
    o.gsl_api = 2
    # (b)
    ts = 1.0e6 + np.array([0.0, 0.05, 0.1])           # 1 ulp of t is 1.2e-10: h shrinks below it long before any
    eps = 1e-30                                        # step could meet this tolerance (roundoff in yerr ~ 1e-17 h)
    st[:] = 0
    L.emu_evolve_ham_eps(P(q), P(p), 3, P(ts), P(qo), P(po), LL(B), ctypes.c_double(eps), ctypes.c_double(eps), 5000, I(st), I(ns))
    oq, op, ons = o.evolve_ham_batch(q, p, ts, eps_abs=eps, eps_rel=eps)
    assert np.all(o.last_fail == 1) and np.all(st == 4), (o.last_fail, st)            # HAMK_ST_UNDERFLOW, nothing else
    # yerr is pure rounding here, different in the two evaluation orders: which of the last rejections can no longer shrink differs by a step or two
    assert np.abs(ns - ons).max() <= 3 and 8 <= ns.min() and ns.max() < 20              # 0.2x per rejection down to 1 ulp of t
    assert relerr(qo, oq) < 1e-4 and np.array_equal(qo[2], qo[1]) and relerr(qo[1], q) < 1e-3


def test_device_rkf45_step_is_fifth_order(emulate, oracle_lib):
    """The device code's Fehlberg tableau by what defines it: one forced step of size h is the
    5th-order solution, its error against a converged reference falls as h^6 (tests/test_rkf45_order.py
    does the same for the oracle -- neither takes the other's word for the coefficients)."""
    spec = E.get("doublePendulum")
    o = oracle_lib.OracleSystem(spec)
    L, _ = emulate(spec)
    B = 16
    q, qd = E.sample_config(spec, 21, B)
    p = o.to_phase_batch(q, qd)
    hs = [0.2 / 2 ** k for k in range(7)]
    errs = []
    for h in hs:
        tq, tp = o.rk4_steps_batch(q, p, h / 4000, 4000)
        q1, p1, st, ns = q.copy(), p.copy(), np.zeros(B, np.int32), np.zeros(B, np.int32)
        L.emu_single_rkf45_step(P(q1), P(p1), LL(B), ctypes.c_double(h), I(st), I(ns))
        assert np.all(ns == 1) and not st.any()
        errs.append(max(np.abs(q1 - tq).max(), np.abs(p1 - tp).max()))
    ratios = [errs[k] / errs[k + 1] for k in range(len(hs) - 1) if 2e-12 < errs[k + 1] and errs[k] < 1e-6]
    assert ratios and all(40 < r < 100 for r in ratios), (errs, ratios)


@pytest.mark.parametrize("name,loop", [("doublePendulum", "0"), ("doublePendulum", "1"), ("twoBody", "0"), ("spring", "0"), ("threeBodyPolar", "1")])
def test_fixed_step_loop_over_many_steps_on_host(emulate, oracle_lib, name, loop):
    """The RK4 loops with the sincos anchor at each step's midpoint, unrolled and stage-loop bodies:
    40 steps against the oracle -- and a step is a pure function of the state: 40 steps in one launch
    and in launches of 25 + 15 are the same bits (no anchor crosses a step)."""
    spec = E.get(name)
    o = oracle_lib.OracleSystem(spec)
    B = 64
    q, qd = E.sample_config(spec, 17, B)
    p = o.to_phase_batch(q, qd)
    oq, op = o.rk4_steps_batch(q, p, spec.dt, 40)
    L, src = emulate(spec, {"HAMK_RK4_LOOP": loop})
    assert ("RK4_STAGE_LOOP = true" in src) == (loop == "1")
    q2, p2, st = q.copy(), p.copy(), np.zeros(B, np.int32)
    L.emu_rk4(P(q2), P(p2), LL(B), ctypes.c_double(spec.dt), 40, I(st))
    assert not st.any()
    err = np.maximum(np.abs(q2 - oq).max(0), np.abs(p2 - op).max(0))
    calm = np.abs(op).max(0) < 50                             # twoBody: members on their way into a close encounter
    assert calm.mean() > 0.8 and float(err[calm].max()) < 2e-11, (name, float(err[calm].max()))
    q3, p3 = q.copy(), p.copy()
    L.emu_rk4(P(q3), P(p3), LL(B), ctypes.c_double(spec.dt), 25, I(st))
    L.emu_rk4(P(q3), P(p3), LL(B), ctypes.c_double(spec.dt), 15, I(st))
    assert np.array_equal(q3, q2) and np.array_equal(p3, p2)


def test_abs_and_signum_opcodes_on_host(emulate, oracle_lib):
    """Opcodes 27 / 28 (Num.abs, Num.signum) through every AD strategy of the device code against the
    oracle, and the oracle's own derivatives of them against central differences (away from the kinks)."""
    spec = E.get("absZoo")
    o = oracle_lib.OracleSystem(spec)
    q = np.array([0.55, -0.7])
    g, h = o.grad_pe(q), 1e-6
    num = np.array([(o.pe(q + h * e) - o.pe(q - h * e)) / (2 * h) for e in np.eye(2)])
    assert np.max(np.abs(g - num)) < 1e-8
    J = o.jacobian(q)
    numJ = np.stack([(o.coords(q + h * e) - o.coords(q - h * e)) / (2 * h) for e in np.eye(2)], axis=1)
    assert np.max(np.abs(J - numJ)) < 1e-8
    for mode in ("H", "D", "R"):
        L, _ = emulate(spec, {"HAMK_AD_MODE": mode})
        check_against_oracle(L, spec, o, B=32, steps=3, dt_ham=0.02)


@pytest.mark.parametrize("name", ALL_GOLDEN_SYSTEMS)
def test_device_code_on_host_matches_oracle(emulate, oracle_lib, name):
    spec = E.get(name)
    L, _ = emulate(spec)
    check_against_oracle(L, spec, oracle_lib.OracleSystem(spec))


@pytest.mark.parametrize("mode", ["H", "D", "R"])
@pytest.mark.parametrize("name", ["opcodeZoo", "spring", "chain4"])
def test_ad_strategies_and_stage_loops_on_host(emulate, oracle_lib, name, mode):
    """HAMK_AD_MODE = H (full second-order jets), D (directional second sweep), R (generated reverse
    sweep), each with the unrolled and the stage-loop stepping bodies."""
    spec = E.get(name)
    o = oracle_lib.OracleSystem(spec)
    for loop in ("0", "1"):
        L, src = emulate(spec, {"HAMK_AD_MODE": mode, "HAMK_RK4_LOOP": loop, "HAMK_RKF_LOOP": loop, "HAMK_WAVE": "0"})
        assert ("MODE_H = true" in src) == (mode == "H") and ("MODE_R = true" in src) == (mode == "R")
        assert ("RK4_STAGE_LOOP = true" in src) == (loop == "1")
        check_against_oracle(L, spec, o, B=24)


@pytest.mark.parametrize("name", ["chain8", "chain12", "chain16"])
def test_mid_size_systems_on_host(emulate, oracle_lib, name):
    spec = E.get(name)
    L, src = emulate(spec)
    assert "MODE_R = true" in src
    check_against_oracle(L, spec, oracle_lib.OracleSystem(spec), B=8, steps=2, tol=1e-10)


MIXED_LANE = ["doublePendulum~mixed", "spring~mixed", "threeBodyPolar~mixed", "chain6~mixed", "chain12~mixed"]


@pytest.mark.parametrize("name", MIXED_LANE)
def test_mixed_sign_inertias_pivot_like_the_reference_on_host(emulate, oracle_lib, name):
    """The reference inverts EVERY K = J^T M J by LU with partial pivoting (hmatrix `inv`, Hamilton.hs:321, :381): a
    system whose inertias are not all positive -- K symmetric, indefinite, invertible -- is a legal input there.  The lane
    kernels' unpivoted LDL^T meets a non-positive pivot on such a K and falls back, per trajectory, to solve_lu (partial
    pivoting, as the oracle's lu_inverse): VALUES against the oracle, tolerance scaled by cond K -- velocities, hamEqs,
    observables, 5 RK4 steps, stepHam with the oracle's sub-step counts.  n = 2, 3, 6, 6, 12."""
    spec = E.get(name)
    o = oracle_lib.OracleSystem(spec)
    L, src = emulate(spec)
    assert "INERTIA_POS = false" in src
    q, _ = E.sample_config(spec, 11, 24)
    indefinite = [np.linalg.eigvalsh(o.jacobian(q[:, i]).T @ np.diag(spec.inertia) @ o.jacobian(q[:, i])).min() < 0 for i in range(24)]
    assert any(indefinite), "the sample must contain indefinite mass matrices"
    check_against_oracle(L, spec, o, B=24, steps=5, tol=1e-10, qd_kick=0.4, dt_ham=2 * spec.dt)


@pytest.mark.parametrize("body", ["stage-loop (parked)", "unrolled"])
def test_adaptive_stepper_with_parked_stage_vectors_on_host(emulate, oracle_lib, body):
    """hamk_device.hpp rkf45_body_parked (the lane kernels' stage-loop body: the default from n = 4) and the unrolled body
    with everything in registers, chain8: an evolveHam time grid under both GSL bindings with the oracle's sub-step counts
    on every trajectory, and `iterate (stepHam dt)` in one launch == the calls one by one, bitwise."""
    spec = E.get("chain8")
    o = oracle_lib.OracleSystem(spec)
    L, src = emulate(spec, {"HAMK_RKF_LOOP": "1" if body.startswith("stage") else "0"})
    assert ("RKF_STAGE_LOOP = true" in src) == body.startswith("stage")
    B = 9
    q, qd = E.sample_config(spec, 3, B)
    qd = qd + 0.4 * np.cos(np.arange(spec.n * B).reshape(spec.n, B))
    p = o.to_phase_batch(q, qd)
    try:
        for api in (2, 1):
            o.gsl_api = api
            L.emu_set_gsl_api(api)
            ts = np.array([0.0, 0.02, 0.05, 0.05, 0.09])
            qo, po = np.zeros((len(ts), spec.n, B)), np.zeros((len(ts), spec.n, B))
            st, ns = np.zeros(B, np.int32), np.zeros(B, np.int32)
            L.emu_evolve_ham(P(q), P(p), len(ts), P(ts), P(qo), P(po), LL(B), I(st), I(ns))
            oq, op, ons = o.evolve_ham_batch(q, p, ts)
            assert np.array_equal(ns, ons) and not st.any() and ns.min() > 4
            assert relerr(qo, oq) < 1e-10 and relerr(po, op) < 1e-10
            q1, p1, tot = q.copy(), p.copy(), np.zeros(B, np.int64)
            for _ in range(3):
                L.emu_step_ham(P(q1), P(p1), LL(B), ctypes.c_double(0.03), I(st), I(ns))
                tot += ns
            q2, p2 = q.copy(), p.copy()
            fq, fp = np.zeros((3, spec.n, B)), np.zeros((3, spec.n, B))
            L.emu_step_ham_iterate(P(q2), P(p2), LL(B), ctypes.c_double(0.03), 3, 1, P(fq), P(fp), I(st), I(ns))
            assert np.array_equal(q1, q2) and np.array_equal(p1, p2) and np.array_equal(tot, ns.astype(np.int64))
            assert np.array_equal(fq[2], q2) and np.array_equal(fp[2], p2)
    finally:
        L.emu_set_gsl_api(2)


@pytest.mark.parametrize("seed", [0, 3, 8, 10, 12, 15])
def test_random_systems_on_host(emulate, oracle_lib, seed):
    """The random expression-tree systems of the GPU suite (incl. seed 8, whose unrolled RKF45 kernel
    was the nondeterministic one on the GPU): on the host the same device code is right -- the
    defect was the GPU compiler's."""
    from test_gpu_random_systems import random_spec
    spec = random_spec(seed)
    o = oracle_lib.OracleSystem(spec)
    for loop in ("0", "1"):
        L, _ = emulate(spec, {"HAMK_RK4_LOOP": loop, "HAMK_RKF_LOOP": loop})
        check_against_oracle(L, spec, o, B=32, start=99, dt_ham=0.02)


POLY_TRIG_SEEDS = [0, 2, 3, 5, 6, 7, 8, 11, 13, 14, 15]      # (the seeds whose random map has full rank: median cond K < 100)


@pytest.mark.parametrize("seed", POLY_TRIG_SEEDS)
def test_random_trigonometric_polynomial_systems_on_host(emulate, oracle_lib, seed):
    """Round 6: random trigonometric-polynomial maps -- the class whose mass matrix and dT/dq the generator derives symbolically -- through
    the emulated lane kernels against the oracle; at least six of the eleven must actually take the symbolic path."""
    from test_gpu_random_systems import poly_trig_spec
    spec = poly_trig_spec(seed)
    o = oracle_lib.OracleSystem(spec)
    L, src = emulate(spec)
    check_against_oracle(L, spec, o, B=32, start=99, dt_ham=0.02, tol=1e-10)
    test_random_trigonometric_polynomial_systems_on_host.symbolic = getattr(test_random_trigonometric_polynomial_systems_on_host, "symbolic", 0) + ("HAS_SYM_K = true" in src)
    if seed == POLY_TRIG_SEEDS[-1]:
        assert test_random_trigonometric_polynomial_systems_on_host.symbolic >= 6


def check_against_golden(L, name, tol0=1e-12):
    """Device code (any mapping) against the independently derived 50-digit fixtures (tests/golden): no oracle in the loop."""
    from conftest import fvec, load_golden
    pts = load_golden(name)["points"]
    B = len(pts)
    q = np.ascontiguousarray(np.stack([fvec(p["q"]) for p in pts], axis=1))
    qd = np.ascontiguousarray(np.stack([fvec(p["qd"]) for p in pts], axis=1))
    p = np.ascontiguousarray(np.stack([fvec(pt["p"]) for pt in pts], axis=1))
    tol = tol0 * np.maximum(1.0, np.array([float(pt["cond_hint"]) for pt in pts]) / 1e3)
    got = np.zeros_like(q)
    L.emu_to_phase(P(q), P(qd), P(got), LL(B))
    assert np.all(np.abs(got - p).max(0) / np.maximum(1.0, np.abs(p).max(0)) <= tol)
    dq, dp, st = np.zeros_like(q), np.zeros_like(q), np.zeros(B, np.int32)
    L.emu_hameqs(P(q), P(p), P(dq), P(dp), LL(B), I(st))
    wdq = np.stack([fvec(pt["dq"]) for pt in pts], axis=1)
    wdp = np.stack([fvec(pt["dp"]) for pt in pts], axis=1)
    assert not st.any()
    assert np.all(np.abs(dq - wdq).max(0) / np.maximum(1.0, np.abs(wdq).max(0)) <= tol)
    assert np.all(np.abs(dp - wdp).max(0) / np.maximum(1.0, np.abs(wdp).max(0)) <= tol)
    v = np.zeros_like(q)
    L.emu_from_phase(P(q), P(p), P(v), LL(B), I(st))
    wv = np.stack([fvec(pt["vel"]) for pt in pts], axis=1)
    assert np.all(np.abs(v - wv).max(0) / np.maximum(1.0, np.abs(wv).max(0)) <= tol)
    ke, pe, h = np.zeros(B), np.zeros(B), np.zeros(B)
    L.emu_observe(P(q), P(p), P(ke), P(pe), P(h), LL(B), I(st))
    want_h = np.array([float(pt["hamiltonian"]) for pt in pts])
    assert np.all(np.abs(h - want_h) / np.maximum(1.0, np.abs(want_h)) <= tol)


@pytest.mark.parametrize("name", ALL_GOLDEN_SYSTEMS + ["chain8", "chain16"])
def test_device_code_on_host_matches_golden_fixtures(emulate, name):
    """The lane kernels' device code against the 50-digit fixtures -- incl. BASELINE config 5's chain8 / chain16, whose
    fixtures come from the chain's closed-form mechanics (oracle/gen_golden.py evaluate_chain_point)."""
    L, _ = emulate(E.get(name))
    check_against_golden(L, name)


@pytest.fixture(scope="module")
def elementary(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emu_elem") / "elem.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-fno-gnu-unique", "-Wno-unknown-pragmas", "-Wno-attributes",
                           "-include", os.path.join(EMU, "hip_shim.hpp"), "-I" + os.path.join(ROOT, "hamilton_amd", "csrc"),
                           "-o", so, os.path.join(EMU, "sincos_driver.cpp")])
    return ctypes.CDLL(so)


def test_own_sincos_accuracy(elementary):
    """sincos_f64 (3-FMA Cody-Waite + power-basis kernels) against 80-bit long double: <= 2e-16 absolute
    over uniform arguments, arguments next to multiples of pi/2 (where reduction cancels) and large
    arguments up to the hand-over to the library path; the anchored incremental form adds nothing."""
    rng = np.random.default_rng(7)
    k = rng.integers(-1000000, 1000000, 200000).astype(np.float64)
    x = np.concatenate([rng.uniform(-10, 10, 400000), rng.uniform(-1.5e6, 1.5e6, 400000),
                        k * (np.pi / 2) + rng.uniform(-1e-6, 1e-6, k.size), np.array([0.0, -0.0, 1e-300, 1e-9, np.pi / 4, -np.pi / 4])])
    s, c = np.zeros_like(x), np.zeros_like(x)
    elementary.emu_sincos(P(x), P(s), P(c), LL(x.size))
    xl = x.astype(np.longdouble)
    es, ec = np.abs(s - np.sin(xl)).max(), np.abs(c - np.cos(xl)).max()
    assert float(es) < 2.0e-16 and float(ec) < 2.0e-16, (float(es), float(ec))
    xa = rng.uniform(-50, 50, 400000)
    d = rng.uniform(-0.4, 0.4, xa.size)               # beyond |d| = 1/4 the routine falls back to the full evaluation
    elementary.emu_sincos_incr(P(xa), P(d), P(s[:xa.size]), P(c[:xa.size]), LL(xa.size))
    xl = (xa + d).astype(np.longdouble)
    es, ec = np.abs(s[:xa.size] - np.sin(xl)).max(), np.abs(c[:xa.size] - np.cos(xl)).max()
    assert float(es) < 3.0e-16 and float(ec) < 3.0e-16, (float(es), float(ec))


def test_lds_table_sincos_accuracy(elementary):
    """sincos_lut (the stepping kernels' sincos: 512-entry table + a rotation by |r| <= pi/512) against
    80-bit long double over the same argument sets as sincos_f64: <= 2.5e-16 absolute."""
    rng = np.random.default_rng(7)
    k = rng.integers(-1000000, 1000000, 200000).astype(np.float64)
    x = np.concatenate([rng.uniform(-10, 10, 400000), rng.uniform(-1.5e6, 1.5e6, 400000),
                        k * (np.pi / 2) + rng.uniform(-1e-6, 1e-6, k.size),
                        np.arange(-2048, 2048) * (2 * np.pi / 512),                       # the table's own nodes ...
                        (np.arange(-2048, 2048) + 0.5) * (2 * np.pi / 512) * (1 + 1e-15),   # ... and the points between them
                        np.array([0.0, -0.0, 1e-300, 1e-9, np.pi / 4, -np.pi / 4, 1.59e6, -1.59e6])])
    s, c = np.zeros_like(x), np.zeros_like(x)
    elementary.emu_sincos_lut(P(x), P(s), P(c), LL(x.size))
    xl = x.astype(np.longdouble)
    es, ec = np.abs(s - np.sin(xl)), np.abs(c - np.cos(xl))
    assert float(es.max()) < 2.5e-16 and float(ec.max()) < 2.5e-16, (float(es.max()), float(ec.max()))
    assert float(np.sqrt(np.mean(es.astype(np.float64) ** 2))) < 6e-17
    big = np.array([1.7e6, -3e9, 1e22, np.inf, np.nan])                                   # the library path
    sb, cb = np.zeros_like(big), np.zeros_like(big)
    elementary.emu_sincos_lut(P(big), P(sb), P(cb), LL(big.size))
    assert np.allclose(sb[:3], np.sin(big[:3]), atol=1e-15) and np.isnan(sb[3:]).all() and np.isnan(cb[3:]).all()


def test_rotation_ranges(elementary):
    """The three rotation kernels over their whole range (and a little beyond: full re-evaluation
    there), against 80-bit long double: wide |delta| < 1/4, narrow < 1/8, short < 1/32."""
    rng = np.random.default_rng(12)
    n = 600_000
    xa = rng.uniform(-50, 50, n)
    s, c = np.zeros(n), np.zeros(n)
    for rid, lim in ((0, 0.25), (1, 0.125), (2, 0.03125)):
        d = rng.uniform(-1.3 * lim, 1.3 * lim, n)
        d[:4] = [lim * (1 - 1e-12), -lim * (1 - 1e-12), lim, -lim]
        elementary.emu_sincos_incr_range(P(xa), P(d), rid, P(s), P(c), LL(n))
        xl = (xa + d).astype(np.longdouble)
        es, ec = float(np.abs(s - np.sin(xl)).max()), float(np.abs(c - np.cos(xl)).max())
        assert es < 3.0e-16 and ec < 3.0e-16, (rid, es, ec)


def test_midpoint_anchored_sincos_accuracy(elementary):
    """The four sincos evaluations of one RK4 step as the fixed-step loops make them (TRIG_DYN in
    hamk_device.hpp): full at y, narrow rotation to the midpoint (new anchor), short and narrow
    rotations from there -- against 80-bit long double over 1e6 steps with stage offsets up to the
    ranges' limits and a little beyond (lanes beyond a range re-evaluate in full): every pair within a
    few 1e-16, i.e. two rotations cost about one ulp."""
    rng = np.random.default_rng(11)
    n = 1_000_000
    x = rng.uniform(-50, 50, n)
    d1 = rng.uniform(-0.14, 0.14, n)
    d2 = rng.uniform(-0.035, 0.035, n)
    d3 = rng.uniform(-0.14, 0.14, n)
    s, c = np.zeros(4 * n), np.zeros(4 * n)
    elementary.emu_sincos_step(P(x), P(d1), P(d2), P(d3), P(s), P(c), LL(n))
    pts = np.stack([x, x + d1, x + d1 + d2, x + d1 + d3], axis=1).reshape(-1).astype(np.longdouble)
    es, ec = np.abs(s - np.sin(pts)).reshape(n, 4), np.abs(c - np.cos(pts)).reshape(n, 4)
    worst = [float(max(es[:, k].max(), ec[:, k].max())) for k in range(4)]
    assert worst[0] < 2.0e-16 and worst[1] < 3.0e-16 and worst[2] < 4.5e-16 and worst[3] < 4.5e-16, worst


def test_reciprocal_and_controller_power(elementary):
    """frcp (rcp + two Newton steps) and rpow_inv (the step-size controller's r^(-1/5), r^(-1/6)) to
    a few ulp over the ranges the kernels feed them; rpow_inv clamps r to [2^-100, 2^100]."""
    rng = np.random.default_rng(8)
    x = np.concatenate([10.0 ** rng.uniform(-30, 30, 200000), -(10.0 ** rng.uniform(-5, 5, 1000))])
    r = np.zeros_like(x)
    elementary.emu_frcp(P(x), P(r), LL(x.size))
    assert float(np.abs(r * x - 1.0).max()) < 5e-16
    y = 10.0 ** rng.uniform(-28, 28, 200000)
    r5, r6 = np.zeros_like(y), np.zeros_like(y)
    elementary.emu_rpow(P(y), P(r5), P(r6), LL(y.size))
    yl = y.astype(np.longdouble)
    assert float(np.abs(r5 * yl ** (np.longdouble(1) / 5) - 1).max()) < 1e-15
    assert float(np.abs(r6 * yl ** (np.longdouble(1) / 6) - 1).max()) < 1e-15
    edge = np.array([1e-300, 2.0 ** -100, 2.0 ** 100, 1e300])
    e5, e6 = np.zeros_like(edge), np.zeros_like(edge)
    elementary.emu_rpow(P(edge), P(e5), P(e6), LL(edge.size))
    assert e5[0] == e5[1] and e5[2] == e5[3] and abs(e5[1] / 2.0 ** 20 - 1) < 1e-15 and abs(e6[2] * 2.0 ** (100 / 6) - 1) < 1e-15


# ---------------------------------------------------------------------------------------------
# the wave-cooperative kernels (n > 16; hamk_wave.hpp): one OS thread per lane, real barriers,
# cross-lane primitives and the f64 MFMA emulated with the measured operand layout
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def emulate_wave(hamk_lib, tmp_path_factory):
    from hamilton_amd import api
    tmp = tmp_path_factory.mktemp("emu_wave")
    cache = {}

    def make(spec, force):
        from hamilton_amd import _abi
        src = api.system_from_spec(spec, {"mapping": _abi.MAP_WAVE}).source      # (17 <= n <= 32 default to the quad kernels)
        assert "hamk_wave.hpp" in src
        key = hashlib.sha1(src.encode()).hexdigest()[:16]
        if key not in cache:
            cpp, so = str(tmp / f"{key}.cpp"), str(tmp / f"{key}.so")
            with open(cpp, "w") as fh:
                fh.write(src + open(os.path.join(EMU, "wave_driver.inc")).read())
            subprocess.check_call(["g++", "-O1", "-std=c++20", "-pthread", "-fPIC", "-shared", "-fno-gnu-unique", "-Wno-unknown-pragmas", "-Wno-attributes",
                                   "-Wno-psabi", "-include", os.path.join(EMU, "wave_shim.hpp"), "-I" + EMU,
                                   "-I" + os.path.join(ROOT, "hamilton_amd", "csrc"), "-o", so, cpp])
            cache[key] = ctypes.CDLL(so)
        return cache[key]
    return make


@pytest.mark.parametrize("name,force,B", [("spring", True, 9), ("opcodeZoo", True, 6), ("threeBodyPolar", True, 5),
                                          ("chain8", True, 5), ("chain17", False, 3), ("chain18", False, 2), ("chain32", False, 2),
                                          ("chain33", False, 1), ("pendulums40", False, 1), ("dense18", False, 2),
                                          pytest.param("chain64", False, 1, marks=pytest.mark.skipif(
                                              not os.environ.get("HAMK_TEST_SLOW"), reason="68 s of emulated lanes; set HAMK_TEST_SLOW=1 (the GPU suite runs chain64 against the oracle)"))])
def test_wave_kernels_on_host_match_oracle(emulate_wave, oracle_lib, name, force, B):
    """Lane = AD direction, K = J^T M J through the (emulated) matrix-core instruction, two-pivot LDL^T
    by LDS column broadcast with the forward substitution riding along, four-wide back substitution,
    group-uniform RKF45 control -- against the oracle, incl. group sizes 16, 32 and 64 (one trajectory per wavefront), padded lanes,
    panels of 16 pivots with the trailing blocks updated by MFMA (two panels at n <= 32, up to four beyond; a second
    panel of ONE pivot at n = 17), odd N (single last pivot), N mod 4 != 0 (scalar head of the back substitution), M mod 4 != 0
    (zero-padded MFMA rows), unequal inertias, and ensembles that do not fill the last block; one trajectory per wavefront with the
    pivot rows read lane by lane (n > 32); matrix-core blocks skipped where the Jacobian's structure makes them zero -- a chain's
    upper blocks, all but one block per four rows of a block-diagonal map (pendulums40), none of a dense map (dense18)."""
    spec = E.get(name)
    L = emulate_wave(spec, force)
    check_against_oracle(L, spec, oracle_lib.OracleSystem(spec), B=B, steps=2, tol=1e-10)


@pytest.mark.parametrize("name,B", [("chain6~mixed", 5), ("chain20~mixed", 2), ("spring~mixed", 6),
                                    pytest.param("chain33~mixed", 1, marks=pytest.mark.skipif(
                                        not os.environ.get("HAMK_TEST_SLOW"), reason="50 s of emulated lanes (one trajectory per wavefront); set HAMK_TEST_SLOW=1"))])
def test_wave_kernels_pivot_where_an_inertia_is_not_positive(emulate_wave, oracle_lib, name, B):
    """hamk_wave.hpp solve_pivoted: LU with partial pivoting, rows distributed over the lanes of a group -- the wave
    kernels' counterpart of the reference's `inv` (Hamilton.hs:321, :381) for systems whose K need not be definite
    (round 3 flagged every lane HAMK_ST_SINGULAR there).  Group sizes 16, 32 and 64, padded lanes, against the oracle
    with cond-scaled tolerances: velocities, hamEqs, observables, RK4 steps, stepHam."""
    spec = E.get(name)
    o = oracle_lib.OracleSystem(spec)
    L = emulate_wave(spec, name in ("chain6~mixed", "spring~mixed"))
    check_against_oracle(L, spec, o, B=B, steps=2, tol=1e-10, qd_kick=0.4, dt_ham=spec.dt)


def test_wave_pivoted_solve_flags_an_exactly_singular_matrix(emulate_wave, oracle_lib):
    """All-zero inertias: K = 0, nothing to pivot on -- the reference raises out of `inv`; every trajectory is flagged."""
    from dataclasses import replace
    spec = replace(E.get("chain5"), name="chain5~zero", inertia=(0.0,) * 10)
    L = emulate_wave(spec, True)
    B = 3
    q, qd = E.sample_config(spec, 0, B)
    p = np.ones_like(q)
    v, st = np.zeros_like(q), np.zeros(B, np.int32)
    L.emu_from_phase(P(q), P(p), P(v), LL(B), I(st))
    assert np.all(st & 1) and np.all(np.isnan(v))


@pytest.mark.parametrize("name,env", [("doublePendulum", None), ("spring", None), ("threeBodyPolar", None),
                                       ("doublePendulum", {"HAMK_RKF_LOOP": "1", "HAMK_TRIG_LUT": "0"})])
def test_iterate_stepham_is_the_calls_one_by_one(emulate, oracle_lib, name, env):
    """hamk_step_ham_iterate's kernel path: `iterate (stepHam dt)` (README.md:150) inside one kernel invocation is
    BIT-identical to the same number of separate stepHam invocations -- every call restarts from h0 = dt/100 with its
    own budget (Hamilton.hs:400-402, :447), dydt_in of a call is the dydt_out the previous one ended with -- the
    sub-step totals add up, the every-k-th frames are the states the separate calls pass through, and the whole
    sequence agrees with the oracle's stepHam applied as often."""
    spec = E.get(name)
    o = oracle_lib.OracleSystem(spec)
    L, _ = emulate(spec, env)
    B, ncalls, every = 24, 9, 3
    dt = 2 * spec.dt
    q, qd = E.sample_config(spec, 5, B)
    p = o.to_phase_batch(q, qd)
    q1, p1 = q.copy(), p.copy()
    st, ns = np.zeros(B, np.int32), np.zeros(B, np.int32)
    tot = np.zeros(B, np.int64)
    frames_q, frames_p = [], []
    for k in range(ncalls):
        L.emu_step_ham(P(q1), P(p1), LL(B), ctypes.c_double(dt), I(st), I(ns))
        tot += ns
        if (k + 1) % every == 0:
            frames_q.append(q1.copy()); frames_p.append(p1.copy())
    q2, p2 = q.copy(), p.copy()
    fq, fp = np.zeros((ncalls // every, spec.n, B)), np.zeros((ncalls // every, spec.n, B))
    st2, ns2 = np.zeros(B, np.int32), np.zeros(B, np.int32)
    L.emu_step_ham_iterate(P(q2), P(p2), LL(B), ctypes.c_double(dt), ncalls, every, P(fq), P(fp), I(st2), I(ns2))
    assert np.array_equal(q1, q2) and np.array_equal(p1, p2)
    assert np.array_equal(tot, ns2.astype(np.int64))
    assert np.array_equal(np.stack(frames_q), fq) and np.array_equal(np.stack(frames_p), fp)
    oq, op = q.copy(), p.copy()
    for _ in range(ncalls):
        oq, op, _ = o.step_ham_batch(oq, op, dt)
    assert relerr(q2, oq) < 1e-9 and relerr(p2, op) < 1e-9


# ---- four lanes per trajectory (hamk_quad.hpp) ----------------------------------------------------------------------
@pytest.fixture(scope="module")
def emulate_quad(hamk_lib, tmp_path_factory):
    from hamilton_amd import _abi, api
    cache = {}
    tmp = tmp_path_factory.mktemp("emuq")

    def make(spec, options=None, defines=()):
        opt = {"mapping": _abi.MAP_QUAD}
        opt.update(options or {})
        src = "".join(f"#define {d}\n" for d in defines) + api.system_from_spec(spec, opt).source
        assert "hamk_quad.hpp" in src and "HAMK_INSTANTIATE_QUAD" in src
        key = hashlib.sha1(src.encode()).hexdigest()[:16]
        if key not in cache:
            cpp, so = str(tmp / f"{key}.cpp"), str(tmp / f"{key}.so")
            with open(cpp, "w") as fh:
                fh.write(src + open(os.path.join(EMU, "quad_driver.inc")).read())
            subprocess.check_call(["g++", "-O1", "-std=c++20", "-pthread", "-fPIC", "-shared", "-fno-gnu-unique", "-Wno-unknown-pragmas", "-Wno-attributes",
                                   "-Wno-psabi", "-include", os.path.join(EMU, "wave_shim.hpp"), "-I" + EMU,
                                   "-I" + os.path.join(ROOT, "hamilton_amd", "csrc"), "-o", so, cpp])
            cache[key] = ctypes.CDLL(so)
        return cache[key]
    return make


def check_quad_against_oracle(L, spec, o, B, steps=2, tol=1e-10, start=11):
    q, qd = E.sample_config(spec, start, B)
    rng = np.random.default_rng(5)
    qd = qd + 0.3 * rng.standard_normal(qd.shape)           # the chains' box has them at rest
    p = o.to_phase_batch(q, qd)
    st = np.zeros(B, np.int32)
    odq, odp, ost = o.hameqs_batch(q, p)
    assert not ost.any()
    dq, dp = np.zeros_like(q), np.zeros_like(q)
    L.emu_hameqs(P(q), P(p), P(dq), P(dp), LL(B), I(st))
    assert not st.any()
    assert relerr(dq, odq) < tol and relerr(dp, odp) < tol, (relerr(dq, odq), relerr(dp, odp))
    v = np.zeros_like(q)
    L.emu_from_phase(P(q), P(p), P(v), LL(B), I(st))
    assert relerr(v, o.from_phase_batch(q, p)[0]) < tol
    ke, pe, h = np.zeros(B), np.zeros(B), np.zeros(B)
    L.emu_observe(P(q), P(p), P(ke), P(pe), P(h), LL(B), I(st))
    oke, ope, oh = o.observe_batch(q, p)
    assert relerr(ke, oke) < tol and relerr(pe, ope) < tol and relerr(h, oh) < tol
    q2, p2 = q.copy(), p.copy()
    L.emu_rk4(P(q2), P(p2), LL(B), ctypes.c_double(spec.dt), steps, I(st))
    oq, op = o.rk4_steps_batch(q, p, spec.dt, steps)
    assert relerr(q2, oq) < 10 * tol and relerr(p2, op) < 10 * tol, (relerr(q2, oq), relerr(p2, op))
    # one launch = two launches, bitwise; the checked entry point takes the same steps and flags nothing at a loose tolerance
    q3, p3 = q.copy(), p.copy()
    L.emu_rk4(P(q3), P(p3), LL(B), ctypes.c_double(spec.dt), 1, I(st))
    L.emu_rk4(P(q3), P(p3), LL(B), ctypes.c_double(spec.dt), steps - 1, I(st))
    assert np.array_equal(q3, q2) and np.array_equal(p3, p2)
    q4, p4 = q.copy(), p.copy()
    L.emu_rk4_checked(P(q4), P(p4), LL(B), ctypes.c_double(spec.dt), steps, ctypes.c_double(0.5), I(st))
    assert np.array_equal(q4, q2) and not st.any()
    L.emu_rk4_checked(P(q4), P(p4), LL(B), ctypes.c_double(spec.dt), steps, ctypes.c_double(1e-300), I(st))
    assert np.all((st == 16) | (st == 0)) and (st == 16).any()      # HAMK_ST_DRIFT wherever H moved at all


@pytest.mark.parametrize("name,B", [("chain32", 17), ("chain20", 3), ("chain18", 5), ("chain17", 2), ("chain8", 19), ("chain5", 4),
                                    ("threeBodyPolar", 6), ("spring", 9), ("opcodeZoo", 7), ("twoBody", 5)])
def test_quad_kernels_on_host_match_oracle(emulate_quad, oracle_lib, name, B):
    """Every lane runs the per-trajectory sweeps with compile-time seeds, the rows of K are dealt out over the four lanes of
    a quad and factorised in registers with (emulated) DPP broadcasts, the reverse sweep gives dT/dq -- against the oracle:
    n a multiple of four and not (identity padding: 17, 18, 5, 6, 3, 2), one to eight rows per lane, sincos pairs shared
    through LDS (angles as inputs) and per lane (opcodeZoo: sites that are not inputs), potentials over cartesian and
    generalized coordinates, unequal inertias, ensembles that do not fill the last wavefront or block."""
    spec = E.get(name)
    L = emulate_quad(spec)
    check_quad_against_oracle(L, spec, oracle_lib.OracleSystem(spec), B=B)
    if name in ("chain20", "chain18", "chain5", "threeBodyPolar", "spring", "opcodeZoo", "twoBody"):
        # ... and the whole surface (momenta, underlyingPos, keC / lagrangian, stepHam with identical sub-step counts)
        check_against_oracle(L, spec, oracle_lib.OracleSystem(spec), B=min(B, 6), steps=2, tol=1e-10)


@pytest.mark.parametrize("name,B", [("dense18", 5), ("dense24", 3), ("denseMixed17", 4)])       # (dense32 -- four tiles -- on the GPU: test_gpu_wave.py)
def test_quad_dense_maps_on_host_match_oracle(emulate_quad, oracle_lib, name, B):
    """Round 6: coordinate maps with a DENSE Jacobian on the four-lane kernels (hamk_quad.hpp assemble_dense) -- K accumulated in
    tiles (one slot group at n <= 20, low and high slot groups against two column blocks at 24), dU/dq of a cartesian potential as J^T dU/dx in a pass
    of its own, identity padding (18, 17), sincos sites that are inputs (denseN) and that are not, a potential over the generalized
    coordinates (denseMixed17) -- against the oracle, through every entry point of the module."""
    spec = E.get(name)
    L = emulate_quad(spec)
    assert "QUAD_DENSE = true" in __import__("hamilton_amd.api", fromlist=["api"]).system_from_spec(spec, {"mapping": __import__("hamilton_amd._abi", fromlist=["_abi"]).MAP_QUAD}).source
    o = oracle_lib.OracleSystem(spec)
    check_quad_against_oracle(L, spec, o, B=B)
    check_against_oracle(L, spec, o, B=min(B, 3), steps=2, tol=1e-10)


def test_quad_flags_a_singular_mass_matrix(emulate_quad, oracle_lib):
    """twoBody at r = 0: K = diag(mu, mu r^2) has a zero pivot -> HAMK_ST_SINGULAR for that trajectory only."""
    spec = E.get("twoBody")
    L = emulate_quad(spec)
    B = 6
    q, qd = E.sample_config(spec, 3, B)
    p = oracle_lib.OracleSystem(spec).to_phase_batch(q, qd)
    q[0, 2] = 0.0
    st = np.zeros(B, np.int32)
    dq, dp = np.zeros_like(q), np.zeros_like(q)
    L.emu_hameqs(P(q), P(p), P(dq), P(dp), LL(B), I(st))
    assert (st[2] & 1) and not np.delete(st, 2).any()


@pytest.mark.parametrize("seed", [0, 3, 8, 10, 12])
def test_quad_random_systems_on_host(emulate_quad, oracle_lib, seed):
    """Random expression-DAG systems (every smooth opcode, shared subexpressions, unequal inertias, cartesian and
    generalized potentials, n = 1 ... 4: one row per lane, padded quads) on the four-lane kernels against the oracle."""
    from test_gpu_random_systems import random_spec
    spec = random_spec(seed)
    o = oracle_lib.OracleSystem(spec)
    L = emulate_quad(spec)
    B = 21
    q, qd = E.sample_config(spec, 99, B)
    p = o.to_phase_batch(q, qd)
    odq, odp, ost = o.hameqs_batch(q, p)
    cond = np.array([np.linalg.cond(o.jacobian(q[:, i]).T @ np.diag(spec.inertia) @ o.jacobian(q[:, i])) for i in range(B)])
    tol = 1e-11 * np.maximum(1.0, cond / 100.0)
    st = np.zeros(B, np.int32)
    dq, dp = np.zeros_like(q), np.zeros_like(q)
    L.emu_hameqs(P(q), P(p), P(dq), P(dp), LL(B), I(st))
    good = (ost == 0) & (st == 0)
    err = np.maximum(np.abs(dq - odq).max(0) / np.maximum(1.0, np.abs(odq).max(0)), np.abs(dp - odp).max(0) / np.maximum(1.0, np.abs(odp).max(0)))
    assert good.mean() > 0.9 and np.all(err[good] <= tol[good]), (seed, float(np.max(err[good] / tol[good])))
    q2, p2 = q.copy(), p.copy()
    L.emu_rk4(P(q2), P(p2), LL(B), ctypes.c_double(spec.dt), 2, I(st))
    oq, op = o.rk4_steps_batch(q, p, spec.dt, 2)
    e2 = np.maximum(np.abs(q2 - oq).max(0), np.abs(p2 - op).max(0)) / np.maximum(1.0, np.abs(op).max(0))
    assert np.all(e2[good] <= 10 * tol[good]), (seed, float(np.max(e2[good] / tol[good])))


@pytest.mark.parametrize("name", ["chain5", "chain21"])
@pytest.mark.parametrize("api", [2, 1])
def test_quad_adaptive_stepper_on_host(emulate_quad, oracle_lib, api, name):
    """evolveHam over a time grid and `iterate (stepHam dt)` on the four-lane kernels under both GSL bindings: sub-step
    counts identical to the oracle's on every trajectory, one launch of k calls == k launches bitwise.  chain21 takes the
    body whose stage vectors wait in LDS and a run-time-indexed private array (HAMK_QUAD_RKF_PARK, n >= 17), chain5 the other."""
    spec = E.get(name)
    o = oracle_lib.OracleSystem(spec)
    o.gsl_api = api
    L = emulate_quad(spec)
    L.emu_set_gsl_api(api)
    try:
        B = 7
        q, qd = E.sample_config(spec, 3, B)
        qd = qd + 0.4 * np.cos(np.arange(spec.n * B).reshape(spec.n, B))
        p = o.to_phase_batch(q, qd)
        ts = np.array([0.0, 0.02, 0.05, 0.05, 0.09])
        qo, po = np.zeros((len(ts), spec.n, B)), np.zeros((len(ts), spec.n, B))
        st, ns = np.zeros(B, np.int32), np.zeros(B, np.int32)
        L.emu_evolve_ham(P(q), P(p), len(ts), P(ts), P(qo), P(po), LL(B), I(st), I(ns))
        oq, op, ons = o.evolve_ham_batch(q, p, ts)
        assert np.array_equal(ns, ons) and not st.any()
        assert relerr(qo, oq) < 1e-9 and relerr(po, op) < 1e-9
        q1, p1 = q.copy(), p.copy()
        tot = np.zeros(B, np.int64)
        for _ in range(4):
            L.emu_step_ham(P(q1), P(p1), LL(B), ctypes.c_double(0.03), I(st), I(ns))
            tot += ns
        q2, p2 = q.copy(), p.copy()
        fq, fp = np.zeros((2, spec.n, B)), np.zeros((2, spec.n, B))
        L.emu_step_ham_iterate(P(q2), P(p2), LL(B), ctypes.c_double(0.03), 4, 2, P(fq), P(fp), I(st), I(ns))
        assert np.array_equal(q1, q2) and np.array_equal(p1, p2) and np.array_equal(tot, ns.astype(np.int64))
        assert np.array_equal(fq[1], q2) and np.array_equal(fp[1], p2)
        if api == 2:                                       # gsl_odeiv2: a decreasing grid integrates backwards
            tb = np.array([0.0, -0.02, -0.05])
            qb, pb = np.zeros((3, spec.n, B)), np.zeros((3, spec.n, B))
            L.emu_evolve_ham(P(q), P(p), 3, P(tb), P(qb), P(pb), LL(B), I(st), I(ns))
            oqb, opb, onb = o.evolve_ham_batch(q, p, tb)
            assert np.array_equal(ns, onb) and ns.min() > 2 and not st.any()
            assert relerr(qb, oqb) < 1e-9 and relerr(pb, opb) < 1e-9
    finally:
        L.emu_set_gsl_api(2)


@pytest.mark.parametrize("name", ["chain8", "chain16", "chain32"])
def test_quad_kernels_on_host_match_the_chain_fixtures(emulate_quad, name):
    """BASELINE config 5 on the four-lane kernels (the default for chain32; chain8 / chain16 below 32 768 trajectories)
    against the closed-form 50-digit fixtures -- no oracle, no shared tape in the loop."""
    check_against_golden(emulate_quad(E.get(name)), name)


@pytest.mark.parametrize("name", ["chain8", "chain32"])
def test_wave_kernels_on_host_match_the_chain_fixtures(emulate_wave, name):
    check_against_golden(emulate_wave(E.get(name), name == "chain8"), name)


def test_device_sampler_draws_the_numpy_samplers_bits(tmp_path):
    """hamk_sample_batch's kernel (hamk_sample.hpp: splitmix64 keyed by seed, GLOBAL trajectory index and field) on the host
    against examples.sample_config, bit for bit -- every benchmark system's box, shards that start anywhere, sizes that do
    not fill a block, another seed.  (The GPU suite repeats it through the C ABI at 2^20.)"""
    so = str(tmp_path / "sample.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-Wno-attributes",
                           "-I" + EMU, "-I" + os.path.join(ROOT, "hamilton_amd", "csrc"), "-o", so, os.path.join(EMU, "sample_driver.cpp")])
    L = ctypes.CDLL(so)
    for name in ("doublePendulum", "twoBody", "spring", "threeBodyPolar", "chain8", "chain32", "chain64"):
        spec = E.get(name)
        box = [np.array([b[k] for b in bx], dtype=np.float64) for bx in (spec.q_box, spec.qd_box) for k in (0, 1)]
        for start, count, seed in ((0, 300, E.SEED), (1 << 20, 77, E.SEED), (123456789012, 513, 7)):
            q, qd = np.zeros((spec.n, count)), np.zeros((spec.n, count))
            L.emu_sample(P(q), P(qd), LL(count), LL(start), ctypes.c_ulonglong(seed), spec.n, P(box[0]), P(box[1]), P(box[2]), P(box[3]))
            wq, wqd = E.sample_config(spec, start, count, seed)
            assert np.array_equal(q, wq) and np.array_equal(qd, wqd), (name, start)


def test_rejected_attempts_restart_from_the_parked_state_on_host(emulate, oracle_lib):
    """n = 13 (y and dydt are the only rows in LDS): across the last right-hand side of an attempt the trial state and the error
    combination wait in THEIR rows, the old y / dydt in scratch (round 5) -- so a REJECTED attempt must bring them back.  A long
    interval from a kicked start makes the controller reject (checked on the oracle's per-attempt trace); sub-step counts and states
    against the oracle, trajectory by trajectory."""
    spec = E.get("chain13")
    o = oracle_lib.OracleSystem(spec)
    L, _ = emulate(spec)
    B = 6
    q, qd = E.sample_config(spec, 5, B)
    qd = qd + 2.5 * np.cos(1.0 + np.arange(spec.n * B, dtype=np.float64).reshape(spec.n, B))
    p = o.to_phase_batch(q, qd)
    dth = 0.25
    rejected = 0
    for i in range(B):
        tr = o.evolve_ham_trace(q[:, i], p[:, i], np.array([0.0, dth]))[3]
        rejected += sum(1 for (_, _, acc) in tr if acc == 0)
    assert rejected >= 3, rejected                           # the path under test is taken
    q3, p3, ns, st = q.copy(), p.copy(), np.zeros(B, np.int32), np.zeros(B, np.int32)
    L.emu_step_ham(P(q3), P(p3), LL(B), ctypes.c_double(dth), I(st), I(ns))
    sq, sp, sns = o.step_ham_batch(q, p, dth)
    assert np.array_equal(ns, sns), (ns, sns)
    assert relerr(q3, sq) < 1e-8 and relerr(p3, sp) < 1e-8
