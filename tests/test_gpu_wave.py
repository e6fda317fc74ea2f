"""GPU: the wave-cooperative path (hamk_wave.hpp; lane = AD direction, LDS-staged J, shuffle
LDL^T) against the CPU oracle -- on the large-n systems it exists for (chain16, chain32:
BASELINE.json config 5) and, forced with HAMK_WAVE=1, on small systems that also have
high-precision fixtures and a lane-path result to compare with."""
import numpy as np
import pytest

from conftest import fvec, load_golden
from hamilton_amd import examples as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(hamk_lib):
    from hamilton_amd import api as _api
    if hamk_lib.hamk_device_count() < 1:
        pytest.fail("no HIP device visible")
    return _api


def record(**kw):
    import json, os
    path = os.environ.get("HAMK_TEST_RECORD")
    if path:
        with open(path, "a") as fh:
            fh.write(json.dumps(kw) + "\n")


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(1.0, np.abs(np.asarray(b)))))


CASES = [("spring", True), ("threeBodyPolar", True), ("chain4", True), ("opcodeZoo", True),
         ("chain8", True), ("chain16", True), ("chain32", False),
         ("chain33", False), ("chain48", False), ("chain64", False),      # n > 32: one trajectory per wavefront (pivot rows through scalar registers)
         ("pendulums40", False),                                          # block-diagonal Jacobian: one matrix-core block per four rows, at a growing offset
         ("dense24", False), ("dense32", False)]                           # dense Jacobians (round 5): no matrix-core block is skipped; dense32's
                                                                            # RK4 kernel is the one-wavefront build (hamk_dispatch.cpp variant_for)


@pytest.mark.parametrize("name,force", CASES)
def test_wave_path_vs_oracle(api, oracle_lib, monkeypatch, name, force):
    spec = E.get(name)
    from hamilton_amd import _abi
    s = api.system_from_spec(spec, {"mapping": _abi.MAP_WAVE})      # through the ABI's options (17 <= n <= 32 default to the quad kernels)
    assert "HAMK_INSTANTIATE_WAVE" in s.source
    o = oracle_lib.OracleSystem(spec)
    for B in (1, 5, 67):                           # tails: not a multiple of the 8/16 trajectories per block
        q, qd = E.sample_config(spec, 31, B)
        if name.startswith("chain"):
            qd = 0.3 * np.cos(np.arange(spec.n * B).reshape(spec.n, B) * 0.7)   # the C5 box has qd = 0
        p = api.momenta(s, api.Config(q, qd))
        op = o.to_phase_batch(q, qd)
        assert relerr(p, op) < 1e-12, (name, B, "momenta", relerr(p, op))
        dq, dp = api.hamEqs(s, api.Phase(q, op))
        odq, odp, _ = o.hameqs_batch(q, op)
        assert not np.any(s.last_status)
        assert relerr(dq, odq) < 1e-10 and relerr(dp, odp) < 1e-10, (name, B, relerr(dq, odq), relerr(dp, odp))
        v, _ = o.from_phase_batch(q, op)
        assert relerr(api.velocities(s, api.Phase(q, op)), v) < 1e-10
        ke, pe_, h = o.observe_batch(q, op)
        assert relerr(api.keP(s, api.Phase(q, op)), ke) < 1e-10
        assert relerr(api.pe(s, q), pe_) < 1e-12
        assert relerr(api.hamiltonian(s, api.Phase(q, op)), h) < 1e-10
        kc, lg = o.observe_config_batch(q, qd)
        assert relerr(api.keC(s, api.Config(q, qd)), kc) < 1e-10
        assert relerr(api.lagrangian(s, api.Config(q, qd)), lg) < 1e-10
        assert relerr(api.underlyingPos(s, q), o.coords_batch(q)) < 1e-12
        ph = api.rk4Steps(spec.dt, 5, s, api.Phase(q, op))
        oq, opp = o.rk4_steps_batch(q, op, spec.dt, 5)
        assert relerr(ph.positions, oq) < 1e-10 and relerr(ph.momenta, opp) < 1e-10, (name, B)
        assert not np.any(s.last_status)


def test_wave_path_matches_golden_and_lane_path(api, monkeypatch):
    spec = E.get("threeBodyPolar")
    lane = api.system_from_spec(spec)
    monkeypatch.setenv("HAMK_WAVE", "1")
    wave = api.system_from_spec(spec)
    pts = load_golden("threeBodyPolar")["points"]
    q = np.stack([fvec(p["q"]) for p in pts], axis=1)
    p = np.stack([fvec(pt["p"]) for pt in pts], axis=1)
    want_dp = np.stack([fvec(pt["dp"]) for pt in pts], axis=1)
    dq_w, dp_w = api.hamEqs(wave, api.Phase(q, p))
    dq_l, dp_l = api.hamEqs(lane, api.Phase(q, p))
    assert relerr(dp_w, want_dp) < 1e-11 and relerr(dp_w, dp_l) < 1e-11 and relerr(dq_w, dq_l) < 1e-11


@pytest.mark.parametrize("name,force", [("spring", True), ("threeBodyPolar", True), ("chain8", True), ("chain20", False), ("chain40", False)])
def test_wave_adaptive_stepper_vs_oracle(api, oracle_lib, monkeypatch, name, force):
    """stepHam / evolveHam on the wave path: GSL-semantics RKF45 with group-uniform control."""
    spec = E.get(name)
    from hamilton_amd import _abi
    s = api.system_from_spec(spec, {"mapping": _abi.MAP_WAVE})
    assert "HAMK_INSTANTIATE_WAVE" in s.source
    o = oracle_lib.OracleSystem(spec)
    B = 37
    q, qd = E.sample_config(spec, 77, B)
    if name.startswith("chain"):
        qd = 0.3 * np.cos(np.arange(spec.n * B).reshape(spec.n, B) * 0.7)
    p = o.to_phase_batch(q, qd)
    dt = 4 * spec.dt
    st = api.stepHam(dt, s, api.Phase(q, p))
    sq, sp, sns = o.step_ham_batch(q, p, dt)
    same = np.asarray(s.last_nsub) == sns
    assert same.mean() >= 0.97, (name, same.mean(), np.asarray(s.last_nsub)[:8], sns[:8])      # B = 37: one flipped lane allowed
    assert relerr(st.positions[:, same], sq[:, same]) < 1e-9 and relerr(st.momenta[:, same], sp[:, same]) < 1e-9
    assert not np.any(s.last_status)
    ts = np.array([0.0, dt, 2.5 * dt])
    rows = api.evolveHam(s, api.Phase(q, p), ts)
    oq, op, _ = o.evolve_ham_batch(q, p, ts)
    np.testing.assert_array_equal(rows[0].positions, q)
    for r in (1, 2):
        assert relerr(rows[r].positions, oq[r]) < 1e-6 and relerr(rows[r].momenta, op[r]) < 1e-6


def test_singular_flag_on_wave_path(api, monkeypatch):
    monkeypatch.setenv("HAMK_WAVE", "1")
    spec = E.chain(4)
    spec.inertia = (0.0,) * 8
    s = api.system_from_spec(spec)
    q, qd = E.sample_config(spec, 0, 3)
    api.hamEqs(s, api.Phase(q, np.ones_like(q)))
    assert np.all(np.asarray(s.last_status) & 1)


def test_lane_path_at_its_upper_size(api, oracle_lib):
    """n = 12 runs one trajectory per lane (the default up to n = 16 for ensembles that fill the chip; pinned here, 70
    trajectories would otherwise go four lanes each): directional jets, stage loops."""
    from hamilton_amd import _abi
    spec = E.get("chain12")
    s = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE})
    assert "HAMK_INSTANTIATE(HamkSys)" in s.source
    o = oracle_lib.OracleSystem(spec)
    q, qd = E.sample_config(spec, 5, 70)
    qd = 0.3 * np.cos(np.arange(spec.n * 70).reshape(spec.n, 70) * 0.7)
    p = o.to_phase_batch(q, qd)
    dq, dp = api.hamEqs(s, api.Phase(q, p))
    odq, odp, _ = o.hameqs_batch(q, p)
    assert relerr(dq, odq) < 1e-10 and relerr(dp, odp) < 1e-10
    ph = api.rk4Steps(spec.dt, 3, s, api.Phase(q, p))
    oq, op = o.rk4_steps_batch(q, p, spec.dt, 3)
    assert relerr(ph.positions, oq) < 1e-10 and relerr(ph.momenta, op) < 1e-10
    st = api.stepHam(0.005, s, api.Phase(q, p))
    sq, sp, _ = o.step_ham_batch(q, p, 0.005)
    assert relerr(st.positions, sq) < 1e-8 and relerr(st.momenta, sp) < 1e-8


# ---- four lanes per trajectory (hamk_quad.hpp): the default for 17 <= n <= 32 with a sparse Jacobian ------------------
QUAD_CASES = [("chain32", False), ("chain20", False), ("chain17", False), ("chain18", False),
              ("chain16", True), ("chain8", True), ("threeBodyPolar", True), ("spring", True), ("opcodeZoo", True),
              # round 6: DENSE Jacobians on this mapping (hamk_quad.hpp assemble_dense: K in tiles) -- chosen by the library for dense18 and
              # denseD24 (distinct coefficients), forced for denseMixed17 (sincos sites that are not inputs, generalized potential)
              ("dense18", False), ("dense24", False), ("dense32", False), ("denseD24", False), ("denseMixed17", True)]


@pytest.mark.parametrize("name,force", QUAD_CASES)
def test_quad_path_vs_oracle(api, oracle_lib, name, force):
    """Every lane runs the sparse per-trajectory sweeps, the rows of K are dealt out over a quad and factorised in
    registers with DPP broadcasts (hamk_quad.hpp).  hamEqs / velocities / energies / RK4 against the oracle at ensemble
    sizes that leave the last wavefront and block partly filled; the entry points the quad module does not provide
    (momenta, stepHam, ...) run on the system's other module through the same handle."""
    from hamilton_amd import _abi
    spec = E.get(name)
    s = api.system_from_spec(spec, {"mapping": _abi.MAP_QUAD} if force else None)
    r = s.options()
    assert r["mapping"] == _abi.MAP_QUAD and r["lanes_per_trajectory"] == 4, r
    assert "HAMK_INSTANTIATE_QUAD" in s.source
    o = oracle_lib.OracleSystem(spec)
    for B in (1, 5, 67, 300):
        q, qd = E.sample_config(spec, 31, B)
        if name.startswith("chain"):
            qd = 0.3 * np.cos(np.arange(spec.n * B).reshape(spec.n, B) * 0.7)   # the C5 box has qd = 0
        p = api.momenta(s, api.Config(q, qd))                                   # (wave / lane module)
        assert relerr(p, o.to_phase_batch(q, qd)) < 1e-12
        dq, dp = api.hamEqs(s, api.Phase(q, p))
        odq, odp, ost = o.hameqs_batch(q, p)
        assert not ost.any() and not np.any(s.last_status)
        e1 = max(relerr(dq, odq), relerr(dp, odp))
        assert e1 < 1e-10, (name, B, e1)
        assert relerr(api.velocities(s, api.Phase(q, p)), o.from_phase_batch(q, p)[0]) < 1e-10
        oke, ope, oh = o.observe_batch(q, p)
        assert relerr(api.hamiltonian(s, api.Phase(q, p)), oh) < 1e-10 and relerr(api.keP(s, api.Phase(q, p)), oke) < 1e-10
        assert relerr(api.pe(s, q), ope) < 1e-12
        ph = api.rk4Steps(spec.dt, 5, s, api.Phase(q, p))
        oq, op = o.rk4_steps_batch(q, p, spec.dt, 5)
        e5 = max(relerr(ph.positions, oq), relerr(ph.momenta, op))
        assert e5 < 1e-10 and not np.any(s.last_status), (name, B, e5)
        two = api.rk4Steps(spec.dt, 3, s, api.rk4Steps(spec.dt, 2, s, api.Phase(q, p)))
        assert np.array_equal(two.positions, ph.positions) and np.array_equal(two.momenta, ph.momenta)   # pure function of the state
        st = api.stepHam(2 * spec.dt, s, api.Phase(q, p))                        # adaptive stepper: the other module
        sq, sp, sns = o.step_ham_batch(q, p, 2 * spec.dt)
        same = np.asarray(s.last_nsub) == sns
        assert same.mean() >= 0.98 and max(relerr(st.positions[:, same], sq[:, same]), relerr(st.momenta[:, same], sp[:, same])) < 1e-9
        record(test="quad_vs_oracle", name=name, B=B, hameqs=e1, rk4_5=e5)


@pytest.mark.parametrize("name,quad", [("chain4", False), ("chain8", False), ("threeBodyPolar", False), ("chain13", False), ("chain24", True)])
def test_parked_adaptive_stepper_takes_the_reference_steps(api, oracle_lib, name, quad):
    """The adaptive stepper whose stage vectors wait in LDS / a run-time-indexed private array -- the lane kernels' stage-loop
    body (from n = 4; the only one since round 4) and the quad kernels' default from n = 17 (hamk_options::rkf_park) --
    against the body that keeps them in registers (lane: the unrolled body; quad: rkf_park OFF) and against the oracle: the
    same sub-step counts on every trajectory, states to roundoff; `iterate (stepHam dt)` in one launch == the calls one by
    one, bitwise."""
    from hamilton_amd import _abi
    spec = E.get(name)
    mp = _abi.MAP_QUAD if quad else _abi.MAP_LANE
    if quad:
        on = api.system_from_spec(spec, {"mapping": mp, "rkf_park": _abi.ON})
        off = api.system_from_spec(spec, {"mapping": mp, "rkf_park": _abi.OFF})
    else:
        on = api.system_from_spec(spec, {"mapping": mp, "rkf_body": _abi.BODY_STAGE_LOOP})
        off = api.system_from_spec(spec, {"mapping": mp, "rkf_body": _abi.BODY_UNROLLED})
    assert on.options()["rkf_park"] == _abi.ON and off.options()["rkf_park"] == _abi.OFF
    assert api.system_from_spec(spec, {"mapping": mp}).options()["rkf_park"] == _abi.ON          # what AUTO picks
    o = oracle_lib.OracleSystem(spec)
    for B in (3, 300):
        q, qd = E.sample_config(spec, 17, B)
        if name.startswith("chain"):
            qd = 0.3 * np.cos(np.arange(spec.n * B).reshape(spec.n, B) * 0.7)
        p = o.to_phase_batch(q, qd)
        dt = 4 * spec.dt
        a = api.stepHam(dt, on, api.Phase(q, p))
        na = np.asarray(on.last_nsub).copy()
        b = api.stepHam(dt, off, api.Phase(q, p))
        nb = np.asarray(off.last_nsub).copy()
        assert np.array_equal(na, nb) and na.min() >= 3 and not np.any(on.last_status)
        d = max(relerr(a.positions, b.positions), relerr(a.momenta, b.momenta))
        assert d < 1e-12, (name, B, d)
        k = min(B, 40)
        sq, sp, sns = o.step_ham_batch(q[:, :k], p[:, :k], dt)
        assert np.array_equal(na[:k], sns) and max(relerr(a.positions[:, :k], sq), relerr(a.momenta[:, :k], sp)) < 1e-9
        one = a
        for _ in range(2):
            one = api.stepHam(dt, on, one)
        it = api.iterateStepHam(dt, 3, on, api.Phase(q, p))
        assert np.array_equal(it.positions, one.positions) and np.array_equal(it.momenta, one.momenta)
        record(test="parked_rkf", name=name, B=B, parked_vs_registers=d, mean_substeps=float(na.mean()))
    on.gsl_api = 1
    o.gsl_api = 1
    try:
        q, qd = E.sample_config(spec, 5, 20)
        if name.startswith("chain"):
            qd = 0.3 * np.cos(np.arange(spec.n * 20).reshape(spec.n, 20) * 0.9)
        p = o.to_phase_batch(q, qd)
        ts = np.array([0.0, 2 * spec.dt, 5 * spec.dt, 5 * spec.dt, 9 * spec.dt])
        rows = api.evolveHam(on, api.Phase(q, p), ts)
        oq, op, ons = o.evolve_ham_batch(q, p, ts)
        assert np.array_equal(np.asarray(on.last_nsub), ons)
        assert max(relerr(np.stack([r.positions for r in rows]), oq), relerr(np.stack([r.momenta for r in rows]), op)) < 1e-9
    finally:
        on.gsl_api = 2


def test_heavy_tapes_leave_the_two_wavefront_rk4_kernel(api):
    """The wave mapping's RK4 kernel is capped for two wavefronts per SIMD; where a system's tape spills by the thousand under
    that cap (dense32: 967 registers, 490 GB of HBM traffic per launch) the library rebuilds it for one -- a chain keeps two.
    (dense32 is asked onto the wave kernels here: since round 6 the library's own choice for it is the four-lane mapping.)"""
    from hamilton_amd import _abi
    assert api.system_from_spec(E.get("dense32"), {"mapping": _abi.MAP_WAVE}).options()["rk4_min_waves"] == 1
    assert api.system_from_spec(E.get("chain48")).options()["rk4_min_waves"] == 2
    assert api.system_from_spec(E.get("dense32"), {"mapping": _abi.MAP_WAVE, "rk4_min_waves": 2}).options()["rk4_min_waves"] == 2      # the host's word stands


def test_dense_jacobians_choose_their_kernels(api):
    """17 <= n <= 32 with a dense Jacobian (round 6).  Where a first-order sweep with compile-time seeds is cheap (<= 4 m n operations:
    x = 2 q + A sin q + B cos q costs 2 m n) the four-lane kernels take the system -- K accumulated in tiles, 2.0 x (dense24) to 4 x
    (denseD32) the wave-cooperative kernels' RK4 rate (profiles/r06_dense_quad_ab.jsonl); a built kernel that spills by the hundred
    sends the system back to the wave kernels (hamk_dispatch.cpp variant_for)."""
    from hamilton_amd import _abi
    assert api.system_from_spec(E.get("chain24")).options()["mapping"] == _abi.MAP_QUAD
    for name in ("dense18", "dense24", "denseD24"):
        s = api.system_from_spec(E.get(name))
        assert s.options()["mapping"] == _abi.MAP_QUAD and "QUAD_DENSE = true" in s.source, name
    assert "QUAD_DENSE = false" in api.system_from_spec(E.get("chain24")).source
    # shared `constant x value` products are written out per use in such a module (a tape that draws its coefficients from a small table)
    assert "hamk::opaque_const(" in api.system_from_spec(E.get("dense24")).source and "hamk::opaque_const(" not in api.system_from_spec(E.get("chain24")).source
    # a stated mapping stands
    assert api.system_from_spec(E.get("dense18"), {"mapping": _abi.MAP_WAVE}).options()["mapping"] == _abi.MAP_WAVE
